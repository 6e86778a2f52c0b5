// TEST-ONLY stand-in for the interface integration/HFNetHIPModel.h implements: the enums and the abstract class of the
// reference's include/Extractors/BaseModel.h:10-54 with the enum value integration/hfnet_slam_hip.patch adds
// (kHFNetHIPModel).  An interface has one spelling: the virtual signatures below must equal the reference's, which
// tests/test_adapter.py::test_adapter_compiles_against_the_patched_reference_header checks against the real, patched header
// wherever the reference tree is present.  Used only by tests/cpp/test_adapter.cpp on boxes without the reference tree.
#ifndef BASEMODEL_H
#define BASEMODEL_H

#include <vector>
#include <opencv2/opencv.hpp>

namespace ORB_SLAM3
{

enum ModelType { kHFNetTFModel, kHFNetRTModel, kHFNetVINOModel, kHFNetHIPModel };

enum ModelDetectionMode { kImageToLocalAndGlobal, kImageToLocal, kImageToLocalAndIntermediate, kIntermediateToGlobal };

class BaseModel
{
public:
    virtual ~BaseModel(void) = default;
    virtual bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors, cv::Mat &globalDescriptors,
                        int nKeypointsNum, float threshold) = 0;
    virtual bool Detect(const cv::Mat &image, std::vector<cv::KeyPoint> &vKeyPoints, cv::Mat &localDescriptors,
                        int nKeypointsNum, float threshold) = 0;
    virtual bool Detect(const cv::Mat &intermediate, cv::Mat &globalDescriptors) = 0;
    virtual bool IsValid(void) = 0;
    virtual ModelType Type(void) = 0;
};

} // namespace ORB_SLAM3

#endif
