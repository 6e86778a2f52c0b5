// Drives integration/HFNetHIPModel.h -- the adapter a maintainer of the reference drops into include/Extractors/ -- compiled
// with -DUSE_HIP against the test-only OpenCV stand-in (tests/cpp/opencv_shim): the three Detect overloads of
// include/Extractors/BaseModel.h:38-54 in all four modes, and the Matcher / KeyFrameDatabase / LocalMapping glue the patch
// (integration/hfnet_slam_hip.patch) calls.  usage: test_adapter <model dir with hfnet.hfw> <in.bin> <out.bin>
//   in.bin : int32 w, h, nfeatures; u8 imageA[h*w]; u8 imageB[h*w]          out.bin: flat arrays, compared with the oracle by pytest
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Extractors/HFNetHIPModel.h"

using namespace ORB_SLAM3;

static void put(FILE* f, const void* p, size_t n) { fwrite(p, 1, n, f); }
static void put_i(FILE* f, int v) { put(f, &v, 4); }

struct DummyKeyFrame { unsigned long mnId; cv::Mat mGlobalDescriptors; };     // what HIPGlobalDatabase<KF> needs of KeyFrame

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* fi = fopen(argv[2], "rb");
    int hdr[3];
    if (!fi || fread(hdr, 4, 3, fi) != 3) return 2;
    const int w = hdr[0], h = hdr[1], nfeat = hdr[2];
    std::vector<uint8_t> bufA((size_t)w * h), bufB((size_t)w * h);
    if (fread(bufA.data(), 1, bufA.size(), fi) != bufA.size() || fread(bufB.data(), 1, bufB.size(), fi) != bufB.size()) return 2;
    fclose(fi);
    const cv::Mat imageA(h, w, CV_8UC1, bufA.data()), imageB(h, w, CV_8UC1, bufB.data());

    // before any model exists the glue reports "no HIP engine" and leaves the outputs empty (the call sites fall back to the CPU)
    std::vector<int32_t> none;
    const bool early = HIPSearchByBoW(cv::Mat(4, 256, CV_32F), cv::Mat(4, 256, CV_32F), 0.6f, none);

    const std::string dir = argv[1];
    HFNetHIPModel mLG(dir, kImageToLocalAndGlobal, cv::Vec4i(1, h, w, 1));
    if (!mLG.IsValid()) return 3;                                           // loud failure without a GPU / weights
    HFNetHIPModel mL(dir, kImageToLocal, cv::Vec4i(1, h, w, 1));
    HFNetHIPModel mLI(dir, kImageToLocalAndIntermediate, cv::Vec4i(1, h, w, 1));
    HFNetHIPModel mIG(dir, kIntermediateToGlobal, cv::Vec4i(1, h / 8, w / 8, 96));
    FILE* fo = fopen(argv[3], "wb");

    std::vector<cv::KeyPoint> kA, kB, kL, kI;
    cv::Mat dA, dB, dL, dI, gA, gB, inter, gI, tmp;
    int flags = 0;
    flags |= mLG.Detect(imageA, kA, dA, gA, nfeat, 0.01f) ? 1 : 0;
    flags |= mLG.Detect(imageB, kB, dB, gB, nfeat, 0.01f) ? 2 : 0;
    flags |= mL.Detect(imageA, kL, dL, nfeat, 0.01f) ? 4 : 0;
    flags |= mLI.Detect(imageA, kI, dI, inter, nfeat, 0.01f) ? 8 : 0;
    flags |= mIG.Detect(inter, gI) ? 16 : 0;
    // overloads that do not fit the mode return false (HFNetTFModelV2.cc:65,81,92)
    flags |= mLG.Detect(imageA, kL, tmp, nfeat, 0.01f) ? 32 : 0;
    flags |= mL.Detect(imageA, kL, tmp, tmp, nfeat, 0.01f) ? 64 : 0;
    flags |= mLG.Detect(inter, tmp) ? 128 : 0;
    flags |= mIG.Detect(imageA, kL, tmp, nfeat, 0.01f) ? 256 : 0;
    flags |= (mLG.Type() == kHFNetHIPModel && mL.IsValid() && mLI.IsValid() && mIG.IsValid()) ? 512 : 0;
    flags |= early ? 1024 : 0;
    flags |= (kA.empty() || (kA[0].angle == 0.f && kA[0].octave == 0 && kA[0].class_id == -1 && kA[0].size == 0.f)) ? 2048 : 0;
    put_i(fo, flags);
    const int nA = (int)kA.size(), nB = (int)kB.size();
    put_i(fo, nA); put_i(fo, nB); put_i(fo, (int)kI.size());
    for (auto& k : kA) { float v[3] = {k.pt.x, k.pt.y, k.response}; put(fo, v, 12); }
    put(fo, dA.ptr<float>(), (size_t)nA * 256 * 4);
    put(fo, gA.ptr<float>(), 4096 * 4);
    put(fo, dB.ptr<float>(), (size_t)nB * 256 * 4);
    put(fo, dL.ptr<float>(), (size_t)nA * 256 * 4);
    put(fo, inter.ptr<float>(), (size_t)(h / 8) * (w / 8) * 96 * 4);
    put(fo, gI.ptr<float>(), 4096 * 4);

    // Matcher glue
    std::vector<int32_t> mBow, mTri;
    const bool okBow = HIPSearchByBoW(dA, dB, 0.6f, mBow), okTri = HIPSearchForTriangulation(dA, dB, 0.75f, mTri);
    put_i(fo, (okBow ? 1 : 0) | (okTri ? 2 : 0));
    put(fo, mBow.data(), mBow.size() * 4); put(fo, mTri.data(), mTri.size() * 4);

    // KeyFrameDatabase glue: add three keyframes, erase one, add it again (slot reuse), scan
    DummyKeyFrame kf[3] = {{10, gA}, {11, gB}, {12, gI}};
    HIPGlobalDatabase<DummyKeyFrame> db(64, 4096);
    for (auto& k : kf) db.Add(&k);
    db.Erase(&kf[1]); db.Add(&kf[1]); db.Erase(&kf[2]);
    std::vector<std::pair<DummyKeyFrame*, float> > vScores; std::vector<DummyKeyFrame*> vCand;
    const bool okDb = db.Query(gA, 0, vScores, vCand);
    put_i(fo, okDb ? 1 : 0); put_i(fo, (int)db.Size()); put_i(fo, (int)vScores.size()); put_i(fo, (int)vCand.size());
    for (auto& s : vScores) { put_i(fo, (int)s.first->mnId); put(fo, &s.second, 4); }
    for (auto* c : vCand) put_i(fo, (int)c->mnId);

    // LocalMapping glue: keyframe A against {B, A} in one call, every third row of A and every fourth of B "has a MapPoint"
    std::vector<uint8_t> fA(nA), fB(nB);
    for (int i = 0; i < nA; ++i) fA[i] = i % 3 == 0;
    for (int i = 0; i < nB; ++i) fB[i] = i % 4 == 0;
    std::vector<std::vector<int32_t> > vv;
    const bool okStore = HIPKeyFrameStore::Get().SearchForTriangulation(100, dA, fA, {101, 100}, {&dB, &dA}, {fB, fA}, 0.75f, vv);
    // ... and once more with other flags: the resident blocks are reused, only the flags travel
    for (int i = 0; i < nA; ++i) fA[i] = i % 2 == 0;
    std::vector<std::vector<int32_t> > vv2;
    const bool okStore2 = HIPKeyFrameStore::Get().SearchForTriangulation(100, dA, fA, {101}, {&dB}, {fB}, 0.75f, vv2);
    put_i(fo, (okStore ? 1 : 0) | (okStore2 ? 2 : 0));
    if (okStore && okStore2) { put(fo, vv[0].data(), (size_t)nA * 4); put(fo, vv[1].data(), (size_t)nA * 4); put(fo, vv2[0].data(), (size_t)nA * 4); }
    // after Tracking::Reset keyframe ids start again at 0 (src/Tracking.cc:3231): the SAME ids now name other descriptor blocks.
    // Once without Clear() (the resident-hit check must notice), once behind it (what the patched KeyFrameDatabase::clear() calls).
    std::vector<std::vector<int32_t> > vv3, vv4;
    const bool okStore3 = HIPKeyFrameStore::Get().SearchForTriangulation(100, dB, fB, {101}, {&dA}, {fA}, 0.75f, vv3);
    HIPKeyFrameStore::Get().Clear();
    const bool okStore4 = HIPKeyFrameStore::Get().SearchForTriangulation(101, dB, fB, {100}, {&dA}, {fA}, 0.75f, vv4);
    put_i(fo, (okStore3 ? 1 : 0) | (okStore4 ? 2 : 0));
    if (okStore3 && okStore4) { put(fo, vv3[0].data(), (size_t)nB * 4); put(fo, vv4[0].data(), (size_t)nB * 4); }
    fclose(fo);
    return 0;
}
