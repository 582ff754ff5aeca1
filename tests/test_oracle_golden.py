"""Pins the CPU oracle (oracle/) against vectors produced by the reference
itself (tools/gen_golden.py, build container).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref, hrnet_ref, hungarian_ref, synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_state_dict_contract(w48_shapes):
    assert len(w48_shapes) == 1810
    sd = synth.make_state_dict(w48_shapes, 0, "W0")
    n_params = sum(v.numel() for k, v in sd.items()
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 63827139          # students.py:208 of the reference


@pytest.mark.parametrize("variant", ["W0", "W1"])
def test_hrnet_small_matches_reference(golden_dir, w48_shapes, variant):
    g = _load(golden_dir, "hrnet_small.npz")
    sd = synth.make_state_dict(w48_shapes, 0, variant)
    x = synth.make_images(1, 128, 192)
    p, r = hrnet_ref.hrnet_forward(sd, x, half=False)
    # fp32: same stock ops in the same order -> expect bit equality here; allow
    # 1e-5 so a different host CPU (other oneDNN kernel) does not fail
    np.testing.assert_allclose(p.numpy(), g[variant + "_fp32_preds"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(r.numpy(), g[variant + "_fp32_refined"], rtol=0, atol=1e-5)
    ph, rh = hrnet_ref.hrnet_forward({"1." + k: v for k, v in sd.items()}, x, half=True)
    gp, gr = g[variant + "_half_preds"].astype(np.float32), g[variant + "_half_refined"].astype(np.float32)
    # half wrapper: bit-equal in the build container
    assert np.abs(ph.numpy() - gp).max() <= 4e-3
    assert np.abs(rh.numpy() - gr).max() <= 4e-3
    assert (ph.numpy() == gp).mean() > 0.98 and (rh.numpy() == gr).mean() > 0.98


def test_hrnet_640_matches_reference(golden_dir, w48_shapes):
    g = _load(golden_dir, "hrnet_640.npz")
    sd = synth.make_state_dict(w48_shapes, 0, "W1")
    ph, rh = hrnet_ref.hrnet_forward(sd, synth.make_images(1, 640, 640), half=True)
    assert ph.shape == (1, 34, 160, 160) and rh.shape == (1, 17, 320, 320)
    assert np.abs(ph.numpy()[:, :, ::8, ::8] - g["preds_s8"].astype(np.float32)).max() <= 8e-3
    assert np.abs(rh.numpy()[:, :, ::8, ::8] - g["refined_s8"].astype(np.float32)).max() <= 8e-3
    assert abs(float(ph.double().abs().sum()) - float(g["preds_abs"])) < 1e-4 * float(g["preds_abs"])
    assert abs(float(rh.double().abs().sum()) - float(g["refined_abs"])) < 1e-4 * float(g["refined_abs"])


def test_bilinear_formula(golden_dir):
    g = _load(golden_dir, "bilinear.npz")
    y = decode_ref.upsample_bilinear(torch.from_numpy(g["x"]), 53, 77).numpy()
    np.testing.assert_array_equal(y, g["y"])
    # the explicit fma formula the HIP kernels implement is bit-equal too
    np.testing.assert_array_equal(decode_ref.bilinear_explicit(g["x"], 53, 77), g["y"])
    x = torch.randn(1, 3, 40, 56, generator=torch.Generator().manual_seed(0))
    np.testing.assert_array_equal(decode_ref.bilinear_explicit(x.numpy(), 160, 224),
                                  decode_ref.upsample_bilinear(x, 160, 224).numpy())


def test_numpy_mean_orders():
    rng = np.random.default_rng(0)
    for n in range(1, 18):
        for _ in range(50):
            t = [(rng.standard_normal(1) * 3).astype(np.float32) for _ in range(n)]
            np.testing.assert_array_equal(decode_ref._mean_tags_f32(t), np.mean(t, axis=0))
            t = [(rng.standard_normal(3) * 3).astype(np.float32) for _ in range(n)]
            np.testing.assert_array_equal(decode_ref._mean_tags_f32(t), np.mean(t, axis=0))
    a = rng.random((17, 4)).astype(np.float32)
    assert decode_ref.pairwise8_sum_f32(np.ascontiguousarray(a[:, 2])) / np.float32(17) == a[:, 2].mean()


def test_hungarian_optimal():
    rng = np.random.default_rng(1)
    for _ in range(200):
        nr, nc = rng.integers(1, 7), rng.integers(1, 7)
        if nr > nc:
            nr, nc = nc, nr
        c = np.round(rng.random((nr, nc)) * 5) * 100 - rng.random((nr, nc))
        pairs = hungarian_ref.munkres_compute(c)
        assert sorted(r for r, _ in pairs) == list(range(nr))
        assert len({q for _, q in pairs}) == nr
        tot = sum(c[r, q] for r, q in pairs)
        assert abs(tot - hungarian_ref.brute_force_cost(c)) < 1e-9
    from scipy.optimize import linear_sum_assignment
    for n in (10, 30):
        c = rng.random((n, n)) * 1000
        pairs = hungarian_ref.munkres_compute(c)
        r, q = linear_sum_assignment(c)
        assert pairs == list(zip(r.tolist(), q.tolist()))      # unique optimum
    # structured costs as group.py:66 builds them (ties are common): optimal total
    for _ in range(50):
        a, g = rng.integers(1, 31), rng.integers(1, 31)
        c = np.round(rng.random((a, g)) * 6) * 100 - rng.random((a, 1))
        if a > g:
            c = np.concatenate((c, np.full((a, a - g), 1e10)), 1)
        pairs = hungarian_ref.munkres_compute(c)
        r, q = linear_sum_assignment(c)
        assert abs(sum(c[i, j] for i, j in pairs) - c[r, q].sum()) < 1e-4


def _decode_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "decode_p*.npz")))


@pytest.mark.parametrize("name", ["p0", "p1", "p3", "p10", "p30", "p3_480", "p5_d2", "p40"])
def test_decode_matches_reference(golden_dir, name):
    g = _load(golden_dir, "decode_%s.npz" % name)
    P, h, w, seed, D = [int(v) for v in g["meta"]]
    det, tag = synth.make_decode_maps(P, h, w, seed=seed, tag_dim=D)
    det_t, tag_t = torch.from_numpy(det), torch.from_numpy(tag)
    hp = decode_ref.HeatmapParserRef()
    tk = hp.top_k(det_t, tag_t)
    np.testing.assert_array_equal(tk["val_k"], g["val_k"])
    live = g["val_k"] > 0.1
    np.testing.assert_array_equal(tk["loc_k"][live], g["loc_k"][live])
    np.testing.assert_array_equal(tk["tag_k"][live], g["tag_k"][live])
    matched = hp.match(**tk)
    np.testing.assert_array_equal(matched[0], g["matched"])
    adjusted = hp.adjust([m.copy() for m in matched], det)
    np.testing.assert_array_equal(adjusted[0], g["adjusted"])
    ans, scores = hp.parse(det_t, tag_t, adjust=True, refine=True)
    np.testing.assert_array_equal(ans[0], g["final"])
    np.testing.assert_array_equal(np.array(scores, np.float32), g["scores"])


@pytest.mark.parametrize("name", ["lowres_p4", "lowres_p2_nonsq"])
def test_decode_lowres_pipeline(golden_dir, name):
    g = _load(golden_dir, "decode_%s.npz" % name)
    P, H, W, oh, ow, seed = [int(v) for v in g["meta"]]
    refined, tags = synth.make_lowres_maps(P, H, W, seed=seed)
    hms = decode_ref.upsample_bilinear(torch.from_numpy(refined), oh, ow)
    aes = decode_ref.upsample_bilinear(torch.from_numpy(tags), oh, ow)
    hp = decode_ref.HeatmapParserRef()
    ans, scores = hp.parse(hms, aes.unsqueeze(-1), adjust=True, refine=True)
    assert len(ans[0]) == P
    np.testing.assert_array_equal(ans[0], g["final"])
    np.testing.assert_array_equal(np.array(scores, np.float32), g["scores"])
