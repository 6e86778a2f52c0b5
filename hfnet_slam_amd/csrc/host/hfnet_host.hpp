// hfnet_host.hpp -- header-only C++ mirror of the reference's front-end classes on top of the C ABI
// (include/hfnet_hip.h).  Same class / method names, argument meaning and error behaviour as
//
//   BaseModel, ModelDetectionMode, InitAllModels/GetModelVec/GetGlobalModel
//                                   include/Extractors/BaseModel.h:10-65, src/Extractors/BaseModel.cc:24-113
//   HFextractor                     include/Extractors/HFextractor.h, src/Extractors/HFextractor.cc:82-284
//   Matcher (brute-force bodies)    include/Matcher.h:36-89, src/Matcher.cc:220-263,561-621,845-889,1893-1900
//   KeyFrameDatabase (scans)        include/KeyFrameDatabase.h:50-69, src/KeyFrameDatabase.cc:75-104,170-197
//
// The reference types come from OpenCV (cv::Mat, cv::KeyPoint), which this image does not have; the
// tiny `Mat` / `KeyPoint` below carry exactly the fields the path reads and writes, so the adapter in
// INTEGRATION.md (built where OpenCV exists) is a field-for-field copy of this file with cv:: types.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/hfnet_hip.h"

namespace HFNET_HIP {

// ---- the cv:: subset ------------------------------------------------------------------------------
struct Point2f { float x = 0, y = 0; };
struct KeyPoint {                      // cv::KeyPoint: pt, response, octave, angle (always 0 here)
    Point2f pt;
    float response = 0, angle = 0;
    int octave = 0;
};
struct Mat {                           // continuous row-major matrix of float or uint8
    int rows = 0, cols = 0, channels = 1;
    bool is_u8 = false;
    size_t step = 0;                   // bytes between rows (u8 images may be ROIs)
    std::shared_ptr<std::vector<unsigned char>> store;
    unsigned char* data = nullptr;
    Mat() = default;
    static Mat zeros_f32(int r, int c, int ch = 1) {
        Mat m; m.rows = r; m.cols = c; m.channels = ch; m.is_u8 = false; m.step = (size_t)c * ch * 4;
        m.store = std::make_shared<std::vector<unsigned char>>((size_t)r * m.step, 0); m.data = m.store->data(); return m;
    }
    static Mat wrap_u8(const uint8_t* p, int r, int c, size_t step) {
        Mat m; m.rows = r; m.cols = c; m.is_u8 = true; m.step = step; m.data = const_cast<uint8_t*>(p); return m;
    }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    bool isContinuous() const { return step == (size_t)cols * channels * (is_u8 ? 1 : 4); }
};

enum ModelType { kHFNetTFModel, kHFNetRTModel, kHFNetVINOModel, kHFNetHIPModel };   // BaseModel.h:10-14 + the new backend
enum ModelDetectionMode { kImageToLocalAndGlobal, kImageToLocal, kImageToLocalAndIntermediate, kIntermediateToGlobal };

// ---- BaseModel (include/Extractors/BaseModel.h:38-54) --------------------------------------------
class BaseModel {
public:
    virtual ~BaseModel() = default;
    virtual bool Detect(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, Mat& globalDescriptors,
                        int nKeypointsNum, float threshold) = 0;
    virtual bool Detect(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, int nKeypointsNum, float threshold) = 0;
    virtual bool Detect(const Mat& intermediate, Mat& globalDescriptors) = 0;
    virtual bool IsValid() = 0;
    virtual ModelType Type() = 0;
};

struct EngineHandle {
    hfnet_engine* h = nullptr;
    EngineHandle(const std::string& weights, int device) {
        if (hfnet_engine_create(device, weights.c_str(), &h) != HFNET_OK) h = nullptr;
    }
    ~EngineHandle() { hfnet_engine_destroy(h); }
};

// The new backend: HFNetHIPModel : BaseModel.  Failures print to stderr and return false, as the
// reference backends do (HFNetTFModelV2.cc:100-109); a failed construction gives IsValid() == false.
class HFNetHIPModel : public BaseModel {
public:
    HFNetHIPModel(std::shared_ptr<EngineHandle> engine, ModelDetectionMode mode, int height, int width, int maxKeypoints = 5000)
        : mEngine(std::move(engine)), mMode(mode), mH(height), mW(width) {
        if (!mEngine || !mEngine->h || hfnet_model_create(mEngine->h, (hfnet_mode)mode, height, width, maxKeypoints, &mModel) != HFNET_OK) {
            std::fprintf(stderr, "Failed to create HFNetHIPModel: %s\n", hfnet_last_error());
            mModel = nullptr;
        }
    }
    ~HFNetHIPModel() override { hfnet_model_destroy(mModel); }

    bool Detect(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, Mat& globalDescriptors, int nKeypointsNum,
                float threshold) override {
        if (mMode != kImageToLocalAndGlobal && mMode != kImageToLocalAndIntermediate) return false;   // HFNetTFModelV2.cc:65
        const int G = hfnet_engine_info(mEngine->h, 4), C = hfnet_engine_info(mEngine->h, 1);
        // global descriptor is a 4096 x 1 column (HFNetTFModelV2.cc:173); the intermediate keeps the
        // reference's transposed header over the same NHWC bytes (HFNetTFModelV2.cc:210-214)
        if (mMode == kImageToLocalAndGlobal) globalDescriptors = Mat::zeros_f32(G, 1);
        else globalDescriptors = Mat::zeros_f32(mW / 8, mH / 8, C);
        return Run(image, vKeyPoints, localDescriptors, globalDescriptors.ptr<float>(), nKeypointsNum, threshold);
    }
    bool Detect(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, int nKeypointsNum, float threshold) override {
        if (mMode != kImageToLocal) return false;                                                        // HFNetTFModelV2.cc:81
        return Run(image, vKeyPoints, localDescriptors, nullptr, nKeypointsNum, threshold);
    }
    bool Detect(const Mat& intermediate, Mat& globalDescriptors) override {
        if (mMode != kIntermediateToGlobal || !mModel) return false;                                     // HFNetTFModelV2.cc:92
        globalDescriptors = Mat::zeros_f32(hfnet_engine_info(mEngine->h, 4), 1);
        if (hfnet_model_detect_global(mModel, intermediate.ptr<float>(), globalDescriptors.ptr<float>()) != HFNET_OK) {
            std::fprintf(stderr, "%s\n", hfnet_last_error());
            return false;
        }
        return true;
    }
    bool IsValid() override { return mModel && hfnet_model_is_valid(mModel); }
    ModelType Type() override { return kHFNetHIPModel; }

private:
    bool Run(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, float* aux, int n, float threshold) {
        if (!mModel || image.empty() || !image.is_u8 || image.rows != mH || image.cols != mW) return false;   // shape check: HFNetTFModelV2.cc:103
        std::vector<hfnet_keypoint> kps((size_t)(n > 0 ? n : 1));
        Mat desc = Mat::zeros_f32(n > 0 ? n : 1, HFNET_DESC_DIM);
        int got = 0;
        if (hfnet_model_detect(mModel, image.ptr<uint8_t>(), (int)image.step, n, threshold, kps.data(), desc.ptr<float>(), aux, &got) != HFNET_OK) {
            std::fprintf(stderr, "%s\n", hfnet_last_error());
            return false;
        }
        vKeyPoints.resize((size_t)got);
        for (int i = 0; i < got; ++i) {
            vKeyPoints[i].pt.x = kps[i].x; vKeyPoints[i].pt.y = kps[i].y; vKeyPoints[i].response = kps[i].response;
            vKeyPoints[i].octave = kps[i].octave; vKeyPoints[i].angle = 0;
        }
        desc.rows = got;                                               // localDescriptors = Mat(n, 256, CV_32F)
        localDescriptors = desc;
        return true;
    }
    std::shared_ptr<EngineHandle> mEngine;
    hfnet_model* mModel = nullptr;
    ModelDetectionMode mMode;
    int mH, mW;
};

// ---- HFextractor (HFextractor.cc:82-284) ---------------------------------------------------------
// One object == the extractor + the per-level models InitAllModels would build for it; the whole
// pyramid runs as one ragged batch on the GPU instead of one cv::parallel_for_ worker per level.
class HFextractor {
public:
    HFextractor(std::shared_ptr<EngineHandle> engine, int width, int height, int nfeatures, float threshold, float scaleFactor, int nlevels)
        : mEngine(std::move(engine)), nfeatures(nfeatures), nlevels(nlevels), mW(width), mH(height) {
        if (!mEngine || !mEngine->h ||
            hfnet_extractor_create(mEngine->h, width, height, nfeatures, threshold, scaleFactor, nlevels, 1, &mExtractor) != HFNET_OK) {
            std::fprintf(stderr, "Failed to create HFextractor: %s\n", hfnet_last_error());
            mExtractor = nullptr;
            return;
        }
        mvScaleFactor.resize(nlevels); mnFeaturesPerLevel.resize(nlevels);
        std::vector<int> lw(nlevels), lh(nlevels);
        hfnet_extractor_tables(mExtractor, mvScaleFactor.data(), mnFeaturesPerLevel.data(), lw.data(), lh.data());
    }
    ~HFextractor() { hfnet_extractor_destroy(mExtractor); }
    // returns the number of keypoints, -1 for an empty / non-8-bit image (HFextractor.cc:145)
    int operator()(const Mat& image, std::vector<KeyPoint>& vKeyPoints, Mat& localDescriptors, Mat& globalDescriptors) {
        if (!mExtractor || image.empty() || !image.is_u8 || image.cols != mW || image.rows != mH) return -1;
        std::vector<hfnet_keypoint> kps((size_t)nfeatures);
        Mat desc = Mat::zeros_f32(nfeatures, HFNET_DESC_DIM);
        globalDescriptors = Mat::zeros_f32(hfnet_engine_info(mEngine->h, 4), 1);
        int n = 0;
        if (hfnet_extractor_extract(mExtractor, image.ptr<uint8_t>(), (int)image.step, kps.data(), desc.ptr<float>(),
                                    globalDescriptors.ptr<float>(), &n, nullptr) != HFNET_OK) {
            std::fprintf(stderr, "Error while detecting keypoints: %s\n", hfnet_last_error());
            return (int)vKeyPoints.size();
        }
        vKeyPoints.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            vKeyPoints[i].pt.x = kps[i].x; vKeyPoints[i].pt.y = kps[i].y; vKeyPoints[i].response = kps[i].response;
            vKeyPoints[i].octave = kps[i].octave; vKeyPoints[i].angle = 0;
        }
        desc.rows = n;
        localDescriptors = desc;
        return n;
    }
    int GetLevels() const { return nlevels; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
    hfnet_extractor* handle() const { return mExtractor; }      // for KeyFrameDescriptorStore::putExtracted
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor;

private:
    std::shared_ptr<EngineHandle> mEngine;
    hfnet_extractor* mExtractor = nullptr;
    int nfeatures, nlevels, mW, mH;
};

// ---- Matcher brute-force bodies (Matcher.cc) --------------------------------------------------------
class Matcher {
public:
    static constexpr float TH_HIGH = 0.75f, TH_LOW = 0.6f;          // Matcher.cc:33-34
    explicit Matcher(std::shared_ptr<EngineHandle> engine) : mEngine(std::move(engine)) {}
    // Matcher.cc:1893-1900
    float DescriptorDistance(const Mat& a, const Mat& b) const {
        float d = 0;
        hfnet_descriptor_distance(mEngine->h, a.ptr<float>(), b.ptr<float>(), a.cols, &d);
        return d;
    }
    // body of SearchByBoW after the MapPoint gather (Matcher.cc:248-260): BFMatcher(NORM_L2, crossCheck)
    // + distance < TH_LOW.  vMatch[i] = row of `train` matched to row i of `query`, -1 if none.
    int SearchByBoW(const Mat& query, const Mat& train, std::vector<int>& vMatch, std::vector<float>& vDist) const {
        vMatch.assign((size_t)query.rows, -1); vDist.assign((size_t)query.rows, 0.f);
        int n = 0;
        if (hfnet_match_search_by_bow(mEngine->h, query.ptr<float>(), query.rows, train.ptr<float>(), train.rows, query.cols, TH_LOW,
                                      vMatch.data(), vDist.data(), &n, 0) != HFNET_OK) { std::fprintf(stderr, "%s\n", hfnet_last_error()); return 0; }
        return n;
    }
    // GEMM + mutual arg-max of SearchForTriangulation (Matcher.cc:845-889); the epipolar tests that
    // follow (Matcher.cc:891-911) stay with the caller.
    int SearchForTriangulation(const Mat& des1, const Mat& des2, std::vector<int>& vMatch12) const {
        vMatch12.assign((size_t)des1.rows, -1);
        int n = 0;
        if (hfnet_match_search_for_triangulation(mEngine->h, des1.ptr<float>(), des1.rows, des2.ptr<float>(), des2.rows, des1.cols, TH_HIGH,
                                                 vMatch12.data(), &n, 0) != HFNET_OK) { std::fprintf(stderr, "%s\n", hfnet_last_error()); return 0; }
        return n;
    }

private:
    std::shared_ptr<EngineHandle> mEngine;
};

// ---- KeyFrameDatabase scans (KeyFrameDatabase.cc) ---------------------------------------------------
// The descriptor matrix lives in HBM; a slot id per keyframe is mirrored on the KeyFrame side.  The
// covisibility accumulation that follows the scan (KeyFrameDatabase.cc:107-166) stays on the CPU.
class KeyFrameDatabase {
public:
    KeyFrameDatabase(std::shared_ptr<EngineHandle> engine, int capacity) : mEngine(std::move(engine)), mCapacity(capacity) {
        if (hfnet_db_create(mEngine->h, capacity, hfnet_engine_info(mEngine->h, 4), &mDb) != HFNET_OK) mDb = nullptr;
    }
    ~KeyFrameDatabase() { hfnet_db_destroy(mDb); }
    void add(int slot, const Mat& globalDescriptor) { if (mDb) hfnet_db_add(mDb, slot, globalDescriptor.ptr<float>()); }   // KeyFrameDatabase::add
    void erase(int slot) { if (mDb) hfnet_db_erase(mDb, slot); }
    void clear() { if (mDb) hfnet_db_clear(mDb); }
    // scan + "score > 0.8 * best" of DetectNBestCandidates (KeyFrameDatabase.cc:86-104); relocalisation variant :178-197
    int DetectCandidates(const Mat& query, bool relocalization, std::vector<int>& slots, std::vector<float>& scores, float* bestScore = nullptr) const {
        slots.assign((size_t)mCapacity, 0); scores.assign((size_t)mCapacity, 0.f);
        int n = 0; float best = 0;
        if (!mDb || hfnet_db_query(mDb, query.ptr<float>(), relocalization ? 1 : 0, slots.data(), scores.data(), &n, &best, nullptr) != HFNET_OK) n = 0;
        slots.resize((size_t)n); scores.resize((size_t)n);
        if (bestScore) *bestScore = best;
        return n;
    }

private:
    std::shared_ptr<EngineHandle> mEngine;
    hfnet_db* mDb = nullptr;
    int mCapacity;
};

// ---- device-resident descriptor blocks of keyframes --------------------------------------------------
// What LocalMapping::CreateNewMapPoints / LoopClosing need around Matcher.cc:808 and :231: the block of a keyframe is put
// once when the keyframe is made (slot = the caller's dense keyframe index), and one call matches it against all its
// neighbours.  matches[p] has one entry per descriptor of first[p] (-1: none).
class KeyFrameDescriptorStore {
public:
    KeyFrameDescriptorStore(std::shared_ptr<EngineHandle> engine, int capacity, int maxRows) : mEngine(std::move(engine)), mMaxRows(maxRows) {
        if (hfnet_store_create(mEngine->h, capacity, maxRows, 256, &mStore) != HFNET_OK) mStore = nullptr;
    }
    ~KeyFrameDescriptorStore() { hfnet_store_destroy(mStore); }
    bool IsValid() const { return mStore != nullptr; }
    bool put(int slot, const Mat& descriptors) { return mStore && hfnet_store_put(mStore, slot, descriptors.ptr<float>(), descriptors.rows) == HFNET_OK; }
    // the block of the frame `extractor` produced last, device to device (no upload)
    bool putExtracted(int slot, const HFextractor& extractor) { return mStore && hfnet_store_put_extracted(mStore, slot, extractor.handle(), 0) == HFNET_OK; }
    int rows(int slot) const { return hfnet_store_rows(mStore, slot); }
    // one byte per row: 1 = the keypoint has a MapPoint (KeyFrame::GetMapPointMatches)
    bool setMapPointFlags(int slot, const std::vector<unsigned char>& flags) { return mStore && hfnet_store_set_flags(mStore, slot, flags.data(), (int)flags.size()) == HFNET_OK; }
    // Matcher::SearchForTriangulation's descriptor stage (Matcher.cc:808-871) for many keyframe pairs at once
    // rows = HFNET_ROWS_UNFLAGGED on both sides is the reference's "keypoints without a MapPoint" gather (:808-834)
    bool SearchForTriangulation(const std::vector<int>& first, const std::vector<int>& second, float thHigh, std::vector<std::vector<int>>& matches,
                                std::vector<int>& nMatches, int rows1 = HFNET_ROWS_ALL, int rows2 = HFNET_ROWS_ALL) const {
        return run(first, second, rows1, rows2, thHigh, matches, nullptr, nMatches, true);
    }
    // Matcher::SearchByBoW's descriptor stage (Matcher.cc:231-291) for many pairs at once
    // queryRows = HFNET_ROWS_FLAGGED is the reference's "keypoints with a MapPoint" gather (:231-246)
    bool SearchByBoW(const std::vector<int>& query, const std::vector<int>& train, float thLow, std::vector<std::vector<int>>& matches,
                     std::vector<std::vector<float>>& distances, std::vector<int>& nMatches, int queryRows = HFNET_ROWS_ALL,
                     int trainRows = HFNET_ROWS_ALL) const {
        return run(query, train, queryRows, trainRows, thLow, matches, &distances, nMatches, false);
    }

private:
    bool run(const std::vector<int>& a, const std::vector<int>& b, int rowsA, int rowsB, float th, std::vector<std::vector<int>>& matches, std::vector<std::vector<float>>* distances,
             std::vector<int>& nMatches, bool triangulation) const {
        const int n = (int)a.size();
        if (!mStore || b.size() != a.size()) return false;
        std::vector<int32_t> m((size_t)n * mMaxRows), cnt((size_t)n);
        std::vector<float> d(triangulation ? 0 : (size_t)n * mMaxRows);
        const int rc = triangulation ? hfnet_store_search_for_triangulation(mStore, n, a.data(), b.data(), rowsA, rowsB, th, m.data(), cnt.data())
                                     : hfnet_store_search_by_bow(mStore, n, a.data(), b.data(), rowsA, rowsB, th, m.data(), d.data(), cnt.data());
        if (rc != HFNET_OK) return false;
        matches.resize((size_t)n); nMatches.assign(cnt.begin(), cnt.end());
        if (distances) distances->resize((size_t)n);
        for (int p = 0; p < n; ++p) {
            const int r = rows(a[p]);
            matches[p].assign(m.begin() + (size_t)p * mMaxRows, m.begin() + (size_t)p * mMaxRows + r);
            if (distances) (*distances)[p].assign(d.begin() + (size_t)p * mMaxRows, d.begin() + (size_t)p * mMaxRows + r);
        }
        return true;
    }
    std::shared_ptr<EngineHandle> mEngine;
    hfnet_store* mStore = nullptr;
    int mMaxRows;
};

// ---- InitAllModels / GetModelVec / GetGlobalModel (BaseModel.cc:24-113) ---------------------------
// Same wiring as the TensorRT backend: level 0 = kImageToLocalAndGlobal, other levels kImageToLocal,
// no separate global model.  The factory reports failure instead of exit(-1).
struct ModelSet {
    std::shared_ptr<EngineHandle> engine;
    std::vector<std::unique_ptr<BaseModel>> models;
};
inline bool InitAllModels(ModelSet& set, const std::string& strModelPath, int width, int height, int nLevels, float scaleFactor, int device = 0) {
    set.engine = std::make_shared<EngineHandle>(strModelPath, device);
    set.models.clear();
    if (!set.engine->h) { std::fprintf(stderr, "Failed to load HFNet model at path: %s (%s)\n", strModelPath.c_str(), hfnet_last_error()); return false; }
    float scale = 1.0f;
    for (int level = 0; level < nLevels; ++level) {
        const int h = (int)lrintf(height * scale), w = (int)lrintf(width * scale);
        std::unique_ptr<BaseModel> m(new HFNetHIPModel(set.engine, level == 0 ? kImageToLocalAndGlobal : kImageToLocal, h, w));
        if (!m->IsValid()) return false;
        set.models.push_back(std::move(m));
        scale /= scaleFactor;
    }
    return true;
}

// GetModelVec / GetGlobalModel (BaseModel.cc:95-113).  No separate global model with this backend (level 0 returns the
// global descriptor itself, as with TensorRT: BaseModel.cc:78-81), so GetGlobalModel is nullptr.
inline std::vector<BaseModel*> GetModelVec(const ModelSet& set) {
    std::vector<BaseModel*> v;
    for (const auto& m : set.models) v.push_back(m.get());
    return v;
}
inline BaseModel* GetGlobalModel(const ModelSet&) { return nullptr; }

// Resampler (BaseModel.h:78-80)
inline void Resampler(const ModelSet& set, const float* data, const float* warp, float* output, const int batch_size, const int data_height,
                      const int data_width, const int data_channels, const int num_sampling_points) {
    if (hfnet_resampler(set.engine->h, data, warp, output, batch_size, data_height, data_width, data_channels, num_sampling_points) != HFNET_OK)
        std::fprintf(stderr, "%s\n", hfnet_last_error());
}

}  // namespace HFNET_HIP
