"""Pin the CPU oracle (oracle/nsc_oracle.c) against outputs of the reference itself.

tests/golden/*.npz were produced by oracle/gen_golden.py, which imports the reference encoder
(/root/reference/src/encoding) in the build container.  Stage by stage:
  project      range_image.py:129   bit-exact, except pixels touched by itemised atan2-ULP points
  interpolate  range_image.py:15    bit-exact
  encode       spectral_encoder.py:160  |d - ref| <= 1e-5*|ref| + 1e-7 (float32 FFT noise floor)
"""
import glob
import os

import numpy as np
import pytest

import nsc_oracle as orc

RTOL, ATOL = 1e-5, 1e-7
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "enc_*.npz")))


def _params(g):
    return orc.default_params(n_elevation=int(g["n_elevation"]),
                              elevation_range=tuple(g["elevation_range"]))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[4:-4] for c in CASES])
def test_project_matches_reference(path):
    g = np.load(path)
    p = _params(g)
    raw, idx, kept = orc.project(g["points"], p, want_idx=True)
    diff_pix = np.nonzero(raw.view(np.uint32).ravel() != g["ref_raw"].ravel())[0]
    edge = np.nonzero(idx != g["np_idx"])[0]
    # the itemised list is reproducible
    assert edge.tolist() == g["edge_pts"].tolist()
    # every differing pixel is explained by an itemised edge point (numpy's non-correctly-rounded
    # float32 arctan2 put that point in the neighbouring pixel)
    touched = set(idx[edge].tolist()) | set(g["np_idx"][edge].tolist())
    assert set(diff_pix.tolist()) <= touched
    # and an edge point really is an atan2 rounding matter: pixel indices are neighbours
    for i in edge:
        a, b = int(idx[i]), int(g["np_idx"][i])
        dr, dc = abs(a // 360 - b // 360), abs(a % 360 - b % 360)
        assert (dr <= 1 and dc == 0) or (dr == 0 and dc in (1, 359)) or (dr <= 1 and dc in (1, 359))
    assert kept == int((g["np_idx"] >= 0).sum())


@pytest.mark.parametrize("name", ["uniform20k", "safe20k", "uniform120k", "adversarial", "wide20k",
                                  "xyz_only", "empty", "single_pt", "e64_ring"])
def test_project_bit_exact_clouds(name, golden_dir):
    """Clouds without bin-edge points: the oracle reproduces the reference image bit for bit."""
    g = np.load(os.path.join(golden_dir, f"enc_{name}.npz"))
    raw = orc.project(g["points"], _params(g))
    assert np.array_equal(raw.view(np.uint32), g["ref_raw"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[4:-4] for c in CASES])
def test_interpolate_bit_exact(path):
    g = np.load(path)
    out = orc.interpolate(g["ref_raw"].view(np.float32))
    assert np.array_equal(out.view(np.uint32), g["ref_interp"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[4:-4] for c in CASES])
def test_descriptor_from_reference_image(path):
    g = np.load(path)
    d = orc.encode_range_image(g["ref_interp"].view(np.float32), _params(g))
    ref = g["ref_desc"]
    assert np.all(np.abs(d - ref) <= RTOL * np.abs(ref) + ATOL)
    assert abs(float(d.sum()) - 1.0) < 1e-5


@pytest.mark.parametrize("name", ["uniform20k", "safe20k", "uniform120k", "adversarial", "wide20k",
                                  "xyz_only", "empty", "single_pt", "e64_ring"])
def test_encode_points_end_to_end(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"enc_{name}.npz"))
    d, raw, itp = orc.encode_points(g["points"], _params(g), want_images=True)
    assert np.array_equal(raw.view(np.uint32), g["ref_raw"])
    assert np.array_equal(itp.view(np.uint32), g["ref_interp"])
    assert np.all(np.abs(d - g["ref_desc"]) <= RTOL * np.abs(g["ref_desc"]) + ATOL)


def test_empty_cloud_is_uniform(golden_dir):
    g = np.load(os.path.join(golden_dir, "enc_empty.npz"))
    d = orc.encode_points(g["points"])
    assert np.array_equal(d, np.full(800, np.float32(1.0) / np.float32(800.0)))
    assert np.array_equal(d, g["ref_desc"])


def test_range_image_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "range_images.npz"))
    p = orc.default_params()
    for img, ref in zip(g["imgs16"], g["desc16"]):
        d = orc.encode_range_image(img, p)
        assert np.all(np.abs(d - ref) <= RTOL * np.abs(ref) + ATOL)
    for img, ref in zip(g["imgs64"], g["desc64"]):       # adaptive_avg_pool2d branch
        d = orc.encode_range_image(img, p)
        assert np.all(np.abs(d - ref) <= RTOL * np.abs(ref) + ATOL)


@pytest.mark.parametrize("alpha", [0.5, 1.0, 2.0, 3.0])
def test_bin_lut(alpha, golden_dir):
    g = np.load(os.path.join(golden_dir, "bin_lut.npz"))
    edges, lut = orc.bin_lut(alpha)
    assert np.array_equal(lut, g[f"lut_{alpha}"])
    assert np.allclose(edges, g[f"edges_{alpha}"], rtol=1e-6, atol=1e-4)


def test_threads_agree():
    from neural_spectral_codec_amd import synth
    pts, off = synth.make_clouds_packed(range(5), 4000)
    a = orc.encode_clouds(pts, off, n_threads=1)
    b = orc.encode_clouds(pts, off, n_threads=3)
    assert np.array_equal(a, b)
    assert np.array_equal(a[2], orc.encode_points(pts[off[2]:off[3]]))


def test_intensity_oracle_matches_reference():
    """nsc_oracle.project_intensity vs the reference's project(keep_intensity=True) (tests/golden/intensity.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intensity.npz"))
    for k in ("c0", "c1", "c2", "c3"):                  # c3: NaN / +inf intensities (np.maximum.at propagates NaN)
        img, inten = orc.project_intensity(g[k + "_pts"])
        assert (img.view(np.uint32) == g[k + "_range"].view(np.uint32)).all(), k
        assert (inten.view(np.uint32) == g[k + "_intensity"].view(np.uint32)).all(), k
        assert np.nanmin(inten) >= 0.0 and (inten > 0).sum() > 1000
    assert np.isnan(g["c3_intensity"]).sum() > 100


def test_nearest_interpolation_oracle_matches_reference(golden_dir):
    """interpolate_range_image(method='nearest') (range_image.py:66-87): numpy restatement vs the reference's outputs."""
    g = np.load(os.path.join(golden_dir, "interp_nearest.npz"))
    for raw, want in zip(g["raw"], g["nearest"]):
        got = orc.interpolate_nearest(raw.view(np.float32))
        assert np.array_equal(got.view(np.uint32), want)
