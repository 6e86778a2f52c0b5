// Exercises the C++ mirror of the reference interface (hfnet_slam_amd/csrc/host/hfnet_host.hpp) the way
// the reference's Examples/Utility/test_extractors.cc / test_match_*_feats.cc drive the original classes:
// InitAllModels -> per-level Detect, HFextractor::operator(), Matcher, KeyFrameDatabase.
// usage: test_host_mirror <weights.hfw> <in.bin> <out.bin>
//   in.bin : int32 w, h, nfeatures, nlevels; u8 image[h*w]
//   out.bin: results as flat float/int arrays (see the writes below); compared with the oracle by pytest.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hfnet_slam_amd/csrc/host/hfnet_host.hpp"

using namespace HFNET_HIP;

static void put(FILE* f, const void* p, size_t n) { fwrite(p, 1, n, f); }

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage\n"); return 2; }
    FILE* fi = fopen(argv[2], "rb");
    if (!fi) return 2;
    int hdr[4];
    if (fread(hdr, 4, 4, fi) != 4) return 2;
    const int w = hdr[0], h = hdr[1], nfeat = hdr[2], nlev = hdr[3];
    std::vector<uint8_t> img((size_t)w * h);
    if (fread(img.data(), 1, img.size(), fi) != img.size()) return 2;
    fclose(fi);

    ModelSet set;
    if (!InitAllModels(set, argv[1], w, h, nlev, 1.2f)) return 3;          // loud failure without a GPU / weights
    const Mat image = Mat::wrap_u8(img.data(), h, w, (size_t)w);
    FILE* fo = fopen(argv[3], "wb");

    // level-0 model, both overload families
    std::vector<KeyPoint> kps;
    Mat local, global;
    bool ok = set.models[0]->Detect(image, kps, local, global, nfeat, 0.01f);
    bool wrong = set.models[0]->Detect(image, kps, local, nfeat, 0.01f);          // 5-arg overload on a LocalAndGlobal model -> false
    int n = (int)kps.size(), flags = (ok ? 1 : 0) | (wrong ? 2 : 0) | (set.models[0]->IsValid() ? 4 : 0);
    put(fo, &flags, 4); put(fo, &n, 4);
    for (auto& k : kps) { float v[3] = {k.pt.x, k.pt.y, k.response}; put(fo, v, 12); }
    put(fo, local.ptr<float>(), (size_t)n * 256 * 4);
    put(fo, global.ptr<float>(), 4096 * 4);

    // HFextractor::operator()
    HFextractor ext(set.engine, w, h, nfeat, 0.01f, 1.2f, nlev);
    std::vector<KeyPoint> ekps;
    Mat edesc, eglob;
    int en = ext(image, ekps, edesc, eglob);
    put(fo, &en, 4);
    for (auto& k : ekps) { float v[4] = {k.pt.x, k.pt.y, k.response, (float)k.octave}; put(fo, v, 16); }
    put(fo, edesc.ptr<float>(), (size_t)(en > 0 ? en : 0) * 256 * 4);
    Mat empty;
    int bad = ext(empty, ekps, edesc, eglob);                                         // -> -1 (HFextractor.cc:145)
    put(fo, &bad, 4);

    // Matcher: match the level-0 descriptors against the extractor's
    Matcher matcher(set.engine);
    std::vector<int> m1, m2;
    std::vector<float> d1;
    int nb = matcher.SearchByBoW(local, edesc, m1, d1);
    int nt = matcher.SearchForTriangulation(local, edesc, m2);
    put(fo, &nb, 4); put(fo, m1.data(), m1.size() * 4); put(fo, d1.data(), d1.size() * 4);
    put(fo, &nt, 4); put(fo, m2.data(), m2.size() * 4);
    float dd = 0;
    if (n > 1) { Mat a = Mat::zeros_f32(1, 256), b = Mat::zeros_f32(1, 256);
                 for (int i = 0; i < 256; ++i) { a.ptr<float>()[i] = local.ptr<float>(0)[i]; b.ptr<float>()[i] = local.ptr<float>(1)[i]; }
                 dd = matcher.DescriptorDistance(a, b); }
    put(fo, &dd, 4);

    // KeyFrameDatabase: the frame's own global descriptor must be its best candidate
    KeyFrameDatabase db(set.engine, 16);
    db.add(3, global);
    db.add(5, eglob);
    std::vector<int> slots; std::vector<float> scores; float best = 0;
    int nc = db.DetectCandidates(global, false, slots, scores, &best);
    put(fo, &nc, 4); put(fo, &best, 4);
    put(fo, slots.data(), slots.size() * 4);

    // KeyFrameDescriptorStore: the same two matches through device-resident blocks
    KeyFrameDescriptorStore store(set.engine, 4, std::max(std::max(n, en), 1));
    // slot 2: the block of the frame `ext` extracted last, taken from its device buffers (the empty-image call above ran nothing)
    int okStore = store.IsValid() && store.put(0, local) && store.putExtracted(2, ext) && store.rows(2) == en;
    std::vector<std::vector<int>> sm, tm; std::vector<std::vector<float>> sd; std::vector<int> sn, tn;
    okStore = okStore && store.SearchByBoW({0, 2}, {2, 0}, 0.6f, sm, sd, sn) && store.SearchForTriangulation({0}, {2}, 0.75f, tm, tn);
    put(fo, &okStore, 4);
    if (okStore) {
        put(fo, sn.data(), 8); put(fo, sm[0].data(), sm[0].size() * 4); put(fo, sd[0].data(), sd[0].size() * 4); put(fo, sm[1].data(), sm[1].size() * 4);
        put(fo, tn.data(), 4); put(fo, tm[0].data(), tm[0].size() * 4);
    }
    fclose(fo);
    return 0;
}
