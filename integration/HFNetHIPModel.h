// HFNetHIPModel.h -- the MI355X backend of HFNet_SLAM as a BaseModel (drop into include/Extractors/ of the reference tree).
//
// This is the file a maintainer adds next to HFNetRTModel.h / HFNetTFModelV2.h; integration/hfnet_slam_hip.patch holds the
// edits to the existing files (enum value, factory branch, Extractor.type string, CMake option).  It is compiled only where
// OpenCV exists (the image this repository is built in has none), so it is NOT part of this repository's build; its
// OpenCV-free twin hfnet_slam_amd/csrc/host/hfnet_host.hpp (same class and method names, same C-ABI calls) is compiled and
// GPU-tested here (tests/cpp/test_host_mirror.cpp).
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

#include <iostream>
#include <mutex>
#include <string>
#include <vector>

#include "Extractors/BaseModel.h"

#ifdef USE_HIP

#include "hfnet_hip.h"

namespace ORB_SLAM3
{

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
