// HFNetHIPModel.h -- the MI355X backend of HFNet_SLAM as a BaseModel (drop into include/Extractors/ of the reference tree).
//
// This is the file a maintainer adds next to HFNetRTModel.h / HFNetTFModelV2.h; integration/hfnet_slam_hip.patch holds the
// edits to the existing files (enum value, factory branch, Extractor.type string, CMake option, and the Matcher /
// KeyFrameDatabase / LocalMapping call sites, which use the helpers at the end of this file).  It needs OpenCV, which the
// image this repository is built in does not have: tests/test_adapter.py compiles it -- this very file, -DUSE_HIP -- against
// a test-only stand-in for the few OpenCV types it touches (tests/cpp/opencv_shim) and runs every entry point on the GPU
// against the oracle (a type-check and a behaviour check of this file, NOT a build of the reference).  Its OpenCV-free twin
// hfnet_slam_amd/csrc/host/hfnet_host.hpp mirrors the whole class family (tests/cpp/test_host_mirror.cpp).
//
// Interface implemented: class BaseModel, include/Extractors/BaseModel.h:38-54 of the reference.
// Behaviour mirrored: HFNetTFModelV2 / HFNetRTModel (src/Extractors/HFNetTFModelV2.cc:62-109, HFNetRTModel.cc:84-137):
//   * a Detect overload that does not fit the model's mode returns false (HFNetTFModelV2.cc:65,81,92);
//   * keypoints: pt = level-resolution pixel, response = score, angle 0, octave 0, every other field as a default-
//     constructed cv::KeyPoint has it (HFNetTFModelV2.cc:122-138 fills a default KeyPoint);
//   * localDescriptors: N x 256 CV_32F; globalDescriptors: 4096 x 1 CV_32F (HFNetTFModelV2.cc:153,173) or, in mode
//     kImageToLocalAndIntermediate, the H/8 x W/8 x 96 map under the transposed header of Tensor2Mat (:210-214).
#ifndef HFNETHIPMODEL_H
#define HFNETHIPMODEL_H

#include <algorithm>
#include <iostream>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "Extractors/BaseModel.h"

#ifdef USE_HIP

#include "hfnet_hip.h"

namespace ORB_SLAM3
{

// Tolerance mode of the HIP back end (include/hfnet_hip.h: engine options scores_bf16x3 / desc_bf16x3 / global_bf16x3 -- the network on the bf16
// matrix pipe with split operands, 1.7 x the frames/s; NMS / top-K exact on the score map the device produced, float outputs within the stated
// tolerances).  Off by default (the exact f32 chains); call before InitAllModels, i.e. before the engine exists -- the options are read when
// models are created.  The reference's own fast back end runs the whole network in FP16 (HFNetRTModel.cc:231).
inline bool& HIPToleranceMode(void)
{
    static bool bTolerance = false;
    return bTolerance;
}

// one engine (weights resident on one GPU) per process, shared by all level models like gvpModels shares the TensorRT runtime
inline hfnet_engine* GetHIPEngine(const std::string &strModelDir = std::string(), int device = 0)
{
    static std::mutex mutex;
    static hfnet_engine* engine = nullptr;
    std::lock_guard<std::mutex> lock(mutex);
    if (!engine && !strModelDir.empty())
    {
        const std::string path = strModelDir + "/hfnet.hfw";      // written by `python -m hfnet_slam_amd.tf_checkpoint`
        if (hfnet_engine_create(device, path.c_str(), &engine) != HFNET_OK)
        {
            std::cerr << "Failed to load HFNet HIP model " << path << ": " << hfnet_last_error() << std::endl;
            engine = nullptr;
        }
        else if (HIPToleranceMode())
        {
            for (const char* name : {"scores_bf16x3", "desc_bf16x3", "global_bf16x3"})
                if (hfnet_engine_set_option(engine, name, 1) != HFNET_OK)
                    std::cerr << "HFNet HIP: option " << name << ": " << hfnet_last_error() << std::endl;
        }
    }
    return engine;
}

class HFNetHIPModel : public BaseModel
{
public:
    HFNetHIPModel(const std::string &strModelDir, ModelDetectionMode mode, const cv::Vec4i inputShape)
        : mMode(mode), mInputShape(inputShape)
    {
        hfnet_engine* engine = GetHIPEngine(strModelDir);
        // shape {1, H, W, C}: the image for the image modes, the H/8 x W/8 x 96 map for kIntermediateToGlobal (BaseModel.cc:37,70)
        if (engine && hfnet_model_create(engine, (hfnet_mode)mode, inputShape(1), inputShape(2), HFNET_MAX_KEYPOINTS, &mModel) != HFNET_OK)
        {
            std::cerr << "Failed to create HFNetHIPModel: " << hfnet_last_error() << std::endl;
            mModel = nullptr;
        }
    }
    virtual ~HFNetHIPModel(void) { hfnet_model_destroy(mModel); }

    bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, cv::Mat &globalDescriptors,
                int nKeypointsNum, float threshold) override
    {
        if (mMode != kImageToLocalAndGlobal && mMode != kImageToLocalAndIntermediate) return false;
        if (mMode == kImageToLocalAndGlobal) globalDescriptors = cv::Mat(4096, 1, CV_32F);
        // (H/8 x W/8 with FLOOR division is exact: the graph crops the image to multiples of 8 first, hf_net.py:188-190, so the
        //  three stride-2 'SAME' layers divide evenly -- hfnet_model_create checks layer_7 against exactly this size)
        else globalDescriptors = cv::Mat(cv::Size(mInputShape(1) / 8, mInputShape(2) / 8), CV_32FC(96));
        return Run(image, vKeyPoints, localDescriptors, globalDescriptors.ptr<float>(), nKeypointsNum, threshold);
    }

    bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors,
                int nKeypointsNum, float threshold) override
    {
        if (mMode != kImageToLocal) return false;
        return Run(image, vKeyPoints, localDescriptors, nullptr, nKeypointsNum, threshold);
    }

    bool Detect(const cv::Mat &intermediate, cv::Mat &globalDescriptors) override
    {
        if (mMode != kIntermediateToGlobal || !mModel || !intermediate.isContinuous()) return false;
        globalDescriptors = cv::Mat(4096, 1, CV_32F);
        return hfnet_model_detect_global(mModel, intermediate.ptr<float>(), globalDescriptors.ptr<float>()) == HFNET_OK;
    }

    bool IsValid(void) override { return mModel != nullptr && hfnet_model_is_valid(mModel); }

    ModelType Type(void) override { return kHFNetHIPModel; }

    hfnet_model* Handle(void) { return mModel; }

private:
    bool Run(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, float* aux,
             int nKeypointsNum, float threshold)
    {
        if (!mModel || image.type() != CV_8UC1 || image.rows != mInputShape(1) || image.cols != mInputShape(2)) return false;
        if (nKeypointsNum < 0 || nKeypointsNum > HFNET_MAX_KEYPOINTS) return false;
        std::vector<hfnet_keypoint> kps(std::max(nKeypointsNum, 1));
        cv::Mat desc(std::max(nKeypointsNum, 1), HFNET_DESC_DIM, CV_32F);
        int n = 0;
        // a non-continuous ROI is fine: rows are image.step bytes apart
        if (hfnet_model_detect(mModel, image.data, (int)image.step, nKeypointsNum, threshold, kps.data(), desc.ptr<float>(), aux, &n) != HFNET_OK)
        {
            std::cerr << "Error while detecting keypoints: " << hfnet_last_error() << std::endl;
            return false;
        }
        cv::KeyPoint keypoint;                       // every field the path does not write keeps its default (size included)
        keypoint.angle = 0;
        vKeyPoints.clear();
        vKeyPoints.reserve(n);
        for (int i = 0; i < n; ++i)
        {
            keypoint.pt.x = kps[i].x;
            keypoint.pt.y = kps[i].y;
            keypoint.response = kps[i].response;
            keypoint.octave = kps[i].octave;
            vKeyPoints.emplace_back(keypoint);
        }
        localDescriptors = desc.rowRange(0, n).clone();
        return true;
    }

    hfnet_model* mModel = nullptr;
    ModelDetectionMode mMode;
    cv::Vec4i mInputShape;
};


// =====================================================================================================================
// Glue for the call sites integration/hfnet_slam_hip.patch edits in Matcher.cc / KeyFrameDatabase.cc / LocalMapping.cc.
// Everything returns false (and leaves its outputs in the "nothing matched" state) when the process has no HIP engine --
// the tree was built with USE_HIP but Extractor.type selects the TensorFlow / TensorRT backend -- or when a call fails,
// and the call sites then fall through to the reference's own CPU code.

// cv::BFMatcher(NORM_L2, crossCheck = true).match(query, train) + (distance < thLow): vMatchQ2T[i] = train row or -1
// (src/Matcher.cc:229-260, 574-618)
inline bool HIPSearchByBoW(const cv::Mat &query, const cv::Mat &train, float thLow, std::vector<int32_t> &vMatchQ2T)
{
    vMatchQ2T.assign(query.rows, -1);
    hfnet_engine* engine = GetHIPEngine();
    if (!engine) return false;
    if (query.rows == 0 || train.rows == 0) return true;
    if (query.type() != CV_32F || train.type() != CV_32F || query.cols != HFNET_DESC_DIM || train.cols != HFNET_DESC_DIM ||
        !query.isContinuous() || !train.isContinuous()) return false;
    std::vector<float> vDist(query.rows);
    int nMatches = 0;
    if (hfnet_match_search_by_bow(engine, query.ptr<float>(), query.rows, train.ptr<float>(), train.rows, HFNET_DESC_DIM, thLow,
                                  vMatchQ2T.data(), vDist.data(), &nMatches, 0) != HFNET_OK)
    {
        std::cerr << "hfnet_match_search_by_bow: " << hfnet_last_error() << std::endl;
        vMatchQ2T.assign(query.rows, -1);
        return false;
    }
    return true;
}

// des1 * des2^T, first maximum above 1 - thHigh^2 / 2 per row, cross-checked per column (src/Matcher.cc:845-889):
// vMatch12[i] = row of d2 or -1
inline bool HIPSearchForTriangulation(const cv::Mat &d1, const cv::Mat &d2, float thHigh, std::vector<int32_t> &vMatch12)
{
    vMatch12.assign(d1.rows, -1);
    hfnet_engine* engine = GetHIPEngine();
    if (!engine) return false;
    if (d1.rows == 0 || d2.rows == 0) return true;
    if (d1.type() != CV_32F || d2.type() != CV_32F || d1.cols != HFNET_DESC_DIM || d2.cols != HFNET_DESC_DIM ||
        !d1.isContinuous() || !d2.isContinuous()) return false;
    int nMatches = 0;
    if (hfnet_match_search_for_triangulation(engine, d1.ptr<float>(), d1.rows, d2.ptr<float>(), d2.rows, HFNET_DESC_DIM, thHigh,
                                             vMatch12.data(), &nMatches, 0) != HFNET_OK)
    {
        std::cerr << "hfnet_match_search_for_triangulation: " << hfnet_last_error() << std::endl;
        vMatch12.assign(d1.rows, -1);
        return false;
    }
    return true;
}

// KeyFrameDatabase's global descriptors resident on the GPU (src/KeyFrameDatabase.cc:36-66 add / erase / clear, :86-104 and
// :178-197 the two scans).  KF is KeyFrame (a template so that this header needs no KeyFrame.h); slots are handed out from
// a free list and mirrored in both directions.  A keyframe whose descriptor is not a continuous 4096 x 1 CV_32F column, or
// a database that outgrows its capacity, switches the object off for good (Valid() == false): the caller's CPU loops take
// over with the std::unordered_set they still maintain.
template <class KF>
class HIPGlobalDatabase
{
public:
    explicit HIPGlobalDatabase(int nCapacity = 65536, int nDim = 4096) : mnCapacity(nCapacity), mnDim(nDim) {}
    ~HIPGlobalDatabase() { hfnet_db_destroy(mpDb); }
    HIPGlobalDatabase(const HIPGlobalDatabase&) = delete;
    HIPGlobalDatabase& operator=(const HIPGlobalDatabase&) = delete;

    bool Valid() { return !mbOff && Ensure(); }

    void Add(KF* pKF)
    {
        if (!Valid() || mSlotOfKF.count(pKF)) return;
        const cv::Mat &g = pKF->mGlobalDescriptors;
        if (g.type() != CV_32F || g.rows * g.cols != mnDim || !g.isContinuous() || (mvFreeSlots.empty() && (int)mvSlotToKF.size() >= mnCapacity))
        {
            mbOff = true;
            return;
        }
        int slot;
        if (!mvFreeSlots.empty()) { slot = mvFreeSlots.back(); mvFreeSlots.pop_back(); }
        else { slot = (int)mvSlotToKF.size(); mvSlotToKF.push_back(nullptr); }
        if (hfnet_db_add(mpDb, slot, g.template ptr<float>()) != HFNET_OK) { mvFreeSlots.push_back(slot); mbOff = true; return; }
        mvSlotToKF[slot] = pKF;
        mSlotOfKF[pKF] = slot;
    }

    void Erase(KF* pKF)
    {
        auto it = mSlotOfKF.find(pKF);
        if (it == mSlotOfKF.end()) return;
        if (mpDb && hfnet_db_erase(mpDb, it->second) != HFNET_OK) mbOff = true;
        mvSlotToKF[it->second] = nullptr;
        mvFreeSlots.push_back(it->second);
        mSlotOfKF.erase(it);
    }

    void Clear()
    {
        if (mpDb && hfnet_db_clear(mpDb) != HFNET_OK) mbOff = true;
        mvSlotToKF.clear(); mvFreeSlots.clear(); mSlotOfKF.clear();
    }

    // score = max(0, 1 - ||query - d||) of EVERY keyframe in the database (the reference stores it on each keyframe, and the
    // covisibility accumulation that follows reads it from keyframes that are not candidates), and the keyframes that pass
    // score > 0.8 * best (mode 0, DetectNBestCandidates) / score > max(0.5, 0.8 * best) (mode 1, DetectRelocalizationCandidates)
    bool Query(const cv::Mat &query, int mode, std::vector<std::pair<KF*, float> > &vScores, std::vector<KF*> &vpCandidates)
    {
        vScores.clear(); vpCandidates.clear();
        if (!Valid() || query.type() != CV_32F || query.rows * query.cols != mnDim || !query.isContinuous()) return false;
        mvCandSlot.resize(mnCapacity); mvCandScore.resize(mnCapacity); mvScoresAll.resize(mnCapacity);
        int nCand = 0;
        float best = 0.f;
        if (hfnet_db_query(mpDb, query.template ptr<float>(), mode, mvCandSlot.data(), mvCandScore.data(), &nCand, &best, mvScoresAll.data()) != HFNET_OK)
        {
            std::cerr << "hfnet_db_query: " << hfnet_last_error() << std::endl;
            return false;
        }
        vScores.reserve(mSlotOfKF.size());
        for (size_t slot = 0; slot < mvSlotToKF.size(); ++slot)
            if (mvSlotToKF[slot]) vScores.emplace_back(mvSlotToKF[slot], mvScoresAll[slot]);
        vpCandidates.reserve(nCand);
        for (int i = 0; i < nCand; ++i) vpCandidates.push_back(mvSlotToKF[mvCandSlot[i]]);
        return true;
    }

    size_t Size() const { return mSlotOfKF.size(); }

private:
    bool Ensure()
    {
        if (mpDb) return true;
        hfnet_engine* engine = GetHIPEngine();
        if (!engine) return false;                  // (not an error: no HIP backend in this process; ask again next time)
        if (hfnet_db_create(engine, mnCapacity, mnDim, &mpDb) != HFNET_OK)
        {
            std::cerr << "hfnet_db_create: " << hfnet_last_error() << std::endl;
            mpDb = nullptr; mbOff = true;
            return false;
        }
        return true;
    }

    hfnet_db* mpDb = nullptr;
    int mnCapacity, mnDim;
    bool mbOff = false;
    std::vector<KF*> mvSlotToKF;
    std::vector<int> mvFreeSlots;
    std::unordered_map<KF*, int> mSlotOfKF;
    std::vector<int32_t> mvCandSlot;
    std::vector<float> mvCandScore, mvScoresAll;
};

// Keyframe descriptor blocks resident on the GPU (hfnet_store_*): a keyframe's N x 256 block is uploaded the first time it
// is matched and found by its id afterwards (KeyFrame::mDescriptors is const, so the copy never goes stale); when all slots
// are taken the least recently used one is replaced.  One process-wide store, like the engine.
class HIPKeyFrameStore
{
public:
    static HIPKeyFrameStore& Get() { static HIPKeyFrameStore store; return store; }

    // SearchForTriangulation of one keyframe against many (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:593-624) in
    // ONE call: rows that already have a MapPoint (vbHasPoint*) are left out on the device, as src/Matcher.cc:808-834 gathers
    // them on the CPU; vvMatch12[k][i] = row of keyframe k's block matched to row i of keyframe 1, or -1, by ORIGINAL row numbers.
    bool SearchForTriangulation(unsigned long nId1, const cv::Mat &desc1, const std::vector<uint8_t> &vbHasPoint1,
                                const std::vector<unsigned long> &vnId2, const std::vector<const cv::Mat*> &vpDesc2,
                                const std::vector<std::vector<uint8_t> > &vvbHasPoint2, float thHigh,
                                std::vector<std::vector<int32_t> > &vvMatch12)
    {
        std::lock_guard<std::mutex> lock(mMutex);
        vvMatch12.clear();
        const int nPairs = (int)vnId2.size();
        if (nPairs == 0) return true;
        if (nPairs + 1 > mnSlots || !Ensure()) return false;
        ++mnClock;
        std::vector<int32_t> vSet1(nPairs), vSet2(nPairs), vCount(nPairs);
        const int slot1 = SlotOf(nId1, desc1, vbHasPoint1);
        if (slot1 < 0) return false;
        for (int k = 0; k < nPairs; ++k)
        {
            vSet1[k] = slot1;
            vSet2[k] = SlotOf(vnId2[k], *vpDesc2[k], vvbHasPoint2[k]);
            if (vSet2[k] < 0) return false;
        }
        std::vector<int32_t> vFlat((size_t)nPairs * mnMaxRows, -1);
        if (hfnet_store_search_for_triangulation(mpStore, nPairs, vSet1.data(), vSet2.data(), HFNET_ROWS_UNFLAGGED, HFNET_ROWS_UNFLAGGED,
                                                 thHigh, vFlat.data(), vCount.data()) != HFNET_OK)
        {
            std::cerr << "hfnet_store_search_for_triangulation: " << hfnet_last_error() << std::endl;
            return false;
        }
        vvMatch12.resize(nPairs);
        for (int k = 0; k < nPairs; ++k)
            vvMatch12[k].assign(vFlat.begin() + (size_t)k * mnMaxRows, vFlat.begin() + (size_t)k * mnMaxRows + desc1.rows);
        return true;
    }

    // forget every resident block (called from KeyFrameDatabase::clear(), i.e. on Tracking::Reset / ResetActiveMap, after which
    // keyframe ids start again at 0: the blocks stay allocated, the next use of an id uploads its descriptors afresh)
    void Clear()
    {
        std::lock_guard<std::mutex> lock(mMutex);
        mSlotOfId.clear();
        std::fill(mvSlotUsed.begin(), mvSlotUsed.end(), 0ull);
    }

private:
    HIPKeyFrameStore() {}
    ~HIPKeyFrameStore() { hfnet_store_destroy(mpStore); }

    bool Ensure()
    {
        if (mpStore) return true;
        hfnet_engine* engine = GetHIPEngine();
        if (!engine || mbOff) return false;
        if (hfnet_store_create(engine, mnSlots, mnMaxRows, HFNET_DESC_DIM, &mpStore) != HFNET_OK)
        {
            std::cerr << "hfnet_store_create: " << hfnet_last_error() << std::endl;
            mpStore = nullptr; mbOff = true;
            return false;
        }
        mvSlotId.assign(mnSlots, 0); mvSlotUsed.assign(mnSlots, 0); mvSlotPtr.assign(mnSlots, nullptr);
        return true;
    }

    // slot of keyframe nId (uploading its block if it is not resident), with its per-row "has a MapPoint" flags refreshed
    int SlotOf(unsigned long nId, const cv::Mat &desc, const std::vector<uint8_t> &vbHasPoint)
    {
        if (desc.type() != CV_32F || desc.cols != HFNET_DESC_DIM || desc.rows > mnMaxRows || !desc.isContinuous() ||
            (int)vbHasPoint.size() < desc.rows) return -1;
        int slot;
        auto it = mSlotOfId.find(nId);
        // a resident hit must still BE this keyframe: Tracking::Reset restarts KeyFrame::nNextId at 0 (src/Tracking.cc:3231), so an
        // id can come back with other descriptors (Clear() below handles the reset; the row count and the address of the
        // immutable descriptor block catch whatever reaches here without one)
        if (it != mSlotOfId.end() && (hfnet_store_rows(mpStore, it->second) != desc.rows || mvSlotPtr[it->second] != (const void*)desc.ptr<float>()))
        {
            mvSlotUsed[it->second] = 0;
            mSlotOfId.erase(it);
            it = mSlotOfId.end();
        }
        if (it != mSlotOfId.end()) slot = it->second;
        else
        {
            slot = 0;
            for (int s = 1; s < mnSlots; ++s) if (mvSlotUsed[s] < mvSlotUsed[slot]) slot = s;     // least recently used (0 = never)
            if (mvSlotUsed[slot] == mnClock) return -1;                                           // every slot is part of this call
            if (mvSlotUsed[slot]) mSlotOfId.erase(mvSlotId[slot]);
            if (hfnet_store_put(mpStore, slot, desc.ptr<float>(), desc.rows) != HFNET_OK) return -1;
            mSlotOfId[nId] = slot; mvSlotId[slot] = nId; mvSlotPtr[slot] = (const void*)desc.ptr<float>();
        }
        mvSlotUsed[slot] = mnClock;
        if (desc.rows > 0 && hfnet_store_set_flags(mpStore, slot, vbHasPoint.data(), desc.rows) != HFNET_OK) return -1;
        return slot;
    }

    std::mutex mMutex;
    hfnet_store* mpStore = nullptr;
    bool mbOff = false;
    const int mnSlots = 256, mnMaxRows = HFNET_MAX_KEYPOINTS;
    unsigned long long mnClock = 0;
    std::vector<unsigned long> mvSlotId;
    std::vector<const void*> mvSlotPtr;          // descriptor block a slot was filled from
    std::vector<unsigned long long> mvSlotUsed;
    std::unordered_map<unsigned long, int> mSlotOfId;
};

} // namespace ORB_SLAM3

#else // USE_HIP

namespace ORB_SLAM3
{

class HFNetHIPModel : public BaseModel
{
public:
    HFNetHIPModel(const std::string &strModelDir, ModelDetectionMode mode, const cv::Vec4i inputShape)
    {
        std::cerr << "You must set USE_HIP in CMakeLists.txt to enable the MI355X backend." << std::endl;
        exit(-1);
    }

    virtual bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, cv::Mat &globalDescriptors,
                        int nKeypointsNum, float threshold) override { return false; }

    virtual bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors,
                        int nKeypointsNum, float threshold) override { return false; }

    virtual bool Detect(const cv::Mat &intermediate, cv::Mat &globalDescriptors) override { return false; }

    bool IsValid(void) override { return false; }

    ModelType Type(void) override { return kHFNetHIPModel; }
};

} // namespace ORB_SLAM3

#endif // USE_HIP

#endif // HFNETHIPMODEL_H
