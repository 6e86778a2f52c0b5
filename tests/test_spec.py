"""Host-side geometry vs the reference's constants and vs the oracle's own tables."""
import numpy as np


def test_channel_plan_matches_survey_appendix_b():
    from hfnet_slam_amd.spec import net_spec
    s = net_spec()
    assert s.stem_out == 24
    assert [b.cout for b in s.blocks] == [16, 24, 24, 24, 48, 96, 48, 48, 48, 48, 72, 72, 72, 120, 120, 120, 240]
    assert [b.expand for b in s.blocks] == [24, 96, 144, 144, 144, 288, 576, 288, 288, 288, 288, 432, 432, 432, 720, 720, 720]
    assert [b.index for b in s.blocks if b.residual] == [4, 9, 10, 11, 13, 14, 16, 17]
    assert s.local_channels == 96 and s.global_channels == 240 and s.vlad_dim == 7680


def test_budget_and_level_sizes():
    from hfnet_slam_amd import spec
    assert spec.features_per_level(1000, 4, 1.2) == [322, 268, 224, 186]
    assert spec.level_sizes(752, 480, 4, 1.2) == [(752, 480), (627, 400), (522, 333), (435, 278)]
    assert spec.level_sizes(512, 512, 4, 1.2) == [(512, 512), (427, 427), (356, 356), (296, 296)]
    assert spec.level_sizes(752, 480, 4, 1.2) == spec.model_level_sizes(752, 480, 4, 1.2)
    assert [spec.cropped(v) for v in (627, 400, 522, 333, 435, 278)] == [624, 400, 520, 328, 432, 272]


def test_oracle_tables_agree_with_spec():
    from hfnet_slam_amd import spec
    from oracle import oracle as O
    for (w, h, nf, nl, sf) in [(752, 480, 1000, 4, 1.2), (512, 512, 850, 4, 1.2), (640, 480, 5000, 8, 1.2), (320, 240, 300, 1, 1.2)]:
        rsf, fpl, lw, lh = O.extractor_tables(nf, nl, sf, w, h)
        assert list(fpl) == spec.features_per_level(nf, nl, sf)
        assert list(zip(lw, lh)) == spec.level_sizes(w, h, nl, sf)
        assert int(fpl.sum()) == nf
        assert np.allclose(rsf, [sf ** i for i in range(nl)], rtol=1e-6)


def test_same_padding_rule():
    from hfnet_slam_amd.spec import same_pad
    assert same_pad(480, 3, 2) == (240, 0, 1)      # even input, stride 2: pad only bottom/right
    assert same_pad(47, 3, 2) == (24, 1, 1)        # odd input: symmetric
    assert same_pad(60, 3, 1) == (60, 1, 1)


def test_weight_container_roundtrip(tmp_path):
    from hfnet_slam_amd import weights
    w = weights.synthetic_weights(3)
    p = str(tmp_path / "w.hfw")
    weights.save(p, w)
    r = weights.load(p)
    assert list(r) == list(w)
    for k in w:
        assert r[k].shape == w[k].shape and np.array_equal(r[k], w[k])
    assert weights.spec_from_tensors(r).depth_multiplier == 0.75
    assert sum(v.size for v in w.values()) == 33074433
