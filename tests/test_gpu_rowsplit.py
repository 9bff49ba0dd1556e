"""GPU tests (-m gpu) of the native row-tile split (gip_rowsplit_* of the C-ABI; BASELINE.json configs[4]): N processor handles in one
process, halo rows by hipMemcpyPeerAsync (device-to-device on the one-GPU box), border components merged on the device.  Every result is
compared with the oracle chain on the FULL frame: remap -> StereoBM (+ L/R check) -> filterSpeckles -> float disparity -> cloud."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


def _kw(nd, block, prefilter_type=1, prefilter_size=9):
    return dict(prefilter_type=prefilter_type, prefilter_size=prefilter_size, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=nd,
                texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)


def _frames(w, h, nd, seed, encoding, calib):
    lr, rr, _, _ = synth.raw_pair(w, h, nd, seed=seed, calib=calib)
    if encoding == "mono8":
        return lr, rr
    lc, rc = synth.colorize(lr, 1), synth.colorize(rr, 2)
    if encoding in ("bgr8", "rgb8"):
        return lc, rc
    return synth.bayer_mosaic(lc, encoding), synth.bayer_mosaic(rc, encoding)


def _oracle_chain(O, lraw, rraw, calib, w, h, kw, encoding, spk_size, spk_diff16):
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    rl = O.remap(O.convert_color(lraw, encoding, "mono8"), *ml)
    rr = O.remap(O.convert_color(rraw, encoding, "mono8"), *mr)
    d16 = O.stereobm(rl, rr, O.bm_params(**kw))
    if spk_size > 0:
        d16 = O.filter_speckles(d16, (kw["min_disparity"] - 1) * 16, spk_size, spk_diff16)
    f32 = O.disparity_to_f32(d16, calib["left"]["P"][2], calib["right"]["P"][2])
    col = O.gray_to_bgr(rl) if encoding == "mono8" else O.remap(O.convert_color(lraw, encoding, "bgr8"), *ml)
    pc = O.pack_pointcloud2(O.reproject_to_3d(f32, O.stereo_q(calib["left"]["P"], calib["right"]["P"]), True), col)
    return d16, f32, pc


def _check(got, exp):
    d16, f32, pc = got
    e16, e32, epc = exp
    assert np.array_equal(d16, e16)
    assert np.array_equal(f32, e32)
    gx, ex = pc[..., :12].copy().view(np.float32), epc[..., :12].copy().view(np.float32)
    assert np.array_equal(np.isnan(gx), np.isnan(ex))
    ok = ~np.isnan(ex)
    assert np.allclose(gx[ok], ex[ok], rtol=2.4e-7, atol=0)      # float xyz: <= 1 ulp (direct vs accumulated column term), as for the batch pipeline
    assert np.array_equal(pc[..., 12:], epc[..., 12:])


@pytest.mark.parametrize("n_tiles,halo_mode,host_threads,gather_mode", [(1, 0, 0, 0), (2, 0, 0, 0), (3, 0, 1, 1), (3, 1, 0, 0), (5, 0, 0, 1), (5, 1, 1, 2)])
def test_rowsplit_native_matches_oracle(gpu, oracle, n_tiles, halo_mode, host_threads, gather_mode):
    """Odd height, block 11, speckle filter on: 1..5 tiles, both halo modes (exchange / every tile rectifies its own halo rows), threaded
    and single-threaded enqueue, border records collected by one kernel or by one copy per tile; three frames back to back through the
    same object (the cross-frame event fences)."""
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h, nd, block = 640, 301, 48, 11
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block)
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), n_tiles, "mono8", None, 120, 2.0, halo_mode=halo_mode, host_threads=host_threads, gather_mode=gather_mode)
    plans = [rs.plan(t) for t in range(n_tiles)]
    assert plans[0]["y0"] == 0 and plans[-1]["y1"] == h and all(a["y1"] == b["y0"] for a, b in zip(plans, plans[1:]))
    assert all(p["e0"] % 2 == 0 and p["e0"] <= p["y0"] and p["e1"] >= p["y1"] for p in plans)
    for seed in (21, 22, 23):
        lraw, rraw = _frames(w, h, nd, seed, "mono8", calib)
        rs.upload(lraw, rraw)
        rs.run()
        got = rs.download()
        exp = _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", 120, 32)
        _check(got, exp)
        assert (exp[0] != -16).mean() > 0.2
    tm = rs.last_timing(0)
    assert tm["total_ms"] > 0 and tm["match_ms"] > 0
    rs.close()


def test_rowsplit_native_border_speckles(gpu, oracle):
    """The merge must matter: with the speckle threshold chosen so that components straddling the tile borders are kept only when their
    parts are added up, per-tile filtering gives a different (wrong) map and the merged result equals the oracle's."""
    import torch
    from ros_gpu_stereo_processor_amd import capi, kernels as K, rowsplit as RS
    w, h, nd, block, n_tiles = 512, 256, 32, 9, 4
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block)
    lraw, rraw = _frames(w, h, nd, 31, "mono8", calib)
    found = False
    for size in (40, 150, 600, 2500):
        rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), n_tiles, "mono8", None, size, 1.0)
        rs.upload(lraw, rraw)
        rs.run()
        got = rs.download()
        exp = _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", size, 16)
        _check(got, exp)
        # what a per-tile filter (no merge) would give
        unfiltered = _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", 0, 0)[0]
        naive = np.concatenate([oracle.filter_speckles(unfiltered[rs.plan(t)["y0"]:rs.plan(t)["y1"]].copy(), -16, size, 16) for t in range(n_tiles)])
        found = found or not np.array_equal(naive, exp[0])
        rs.close()
    assert found, "no component straddled a border at any threshold: the test frame does not exercise the merge"


@pytest.mark.parametrize("encoding,prefilter_type", [("bayer_rggb8", 1), ("bgr8", 1), ("rgb8", 1), ("mono8", 0)])
def test_rowsplit_native_encodings_and_normalized_response(gpu, oracle, encoding, prefilter_type):
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h, nd, block = 480, 200, 32, 9
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block, prefilter_type, 7)
    lraw, rraw = _frames(w, h, nd, 41, encoding, calib)
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), 3, encoding, None, 80, 2.0)
    rs.upload(lraw, rraw)
    rs.run()
    got = rs.download()
    _check(got, _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, encoding, 80, 32))
    rs.close()


def test_rowsplit_native_serial_profile_and_device_inputs(gpu, oracle):
    """gip_rowsplit_run_serial (each tile's phases with the device to itself: the per-rank non-communication time) gives the same frame
    as the concurrent run; inputs handed over as device pointers."""
    import torch
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h, nd, block = 640, 240, 64, 15
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block)
    lraw, rraw = _frames(w, h, nd, 51, "mono8", calib)
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), 4, "mono8", None, 100, 2.0)
    lt, rt = torch.from_numpy(lraw).to(gpu), torch.from_numpy(rraw).to(gpu)
    for t in range(4):
        rs.set_inputs(t, lt, rt)
    rs.run()
    a = rs.download()
    tms = rs.run_serial()
    b = rs.download()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    _check(a, _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", 100, 32))
    assert len(tms) == 4 and all(t["match_ms"] > 0 and t["post_ms"] > 0 and abs(t["total_ms"] - (t["front_ms"] + t["halo_ms"] + t["match_ms"] + t["gather_ms"] + t["post_ms"])) < 1e-3
                                  for t in tms)
    rs.close()


def test_rowsplit_native_rejects_bad_configs(gpu):
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h = 320, 100
    calib = synth.scaled_calibration(w, h)
    with pytest.raises(capi.GipError):
        RS.RowSplit(w, h, calib, capi.BMParams(**_kw(32, 21)), 8, "mono8")       # 12-row tiles are thinner than the 11 + 2 rows they must lend
    with pytest.raises(capi.GipError):
        RS.RowSplit(w, h, calib, capi.BMParams(**_kw(32, 9)), 0, "mono8")
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**_kw(32, 9)), 2, "mono8")
    with pytest.raises(capi.GipError):
        rs.run()                                                                   # no input frame
    rs.close()


def test_config5_4k_rowsplit_four_handles_matches_oracle(gpu, oracle):
    """BASELINE.json configs[4] at full size through the C-ABI: 4096x2160 mono8, ndisp 192, block 15, speckle (800, 5), FOUR handles on
    one device, compared bit for bit with the oracle chain on the whole frame (about ten seconds of CPU)."""
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h, nd, block = 4096, 2160, 192, 15
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block)
    lraw, rraw = _frames(w, h, nd, 100, "mono8", calib)
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), 4, "mono8", None, 800, 5.0)
    assert [rs.plan(t)["y1"] - rs.plan(t)["y0"] for t in range(4)] == [540] * 4
    rs.upload(lraw, rraw)
    rs.run()
    got = rs.download()
    rs.close()
    exp = _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", 800, 80)
    _check(got, exp)
    assert (exp[0] != -16).mean() > 0.2


@pytest.mark.parametrize("nd,block,mind,uniq,d12,tex,spk,outputs_pc", [(32, 9, -8, 0, -1, 0, 0, True), (48, 19, 4, 15, 0, 3, 40, True), (64, 5, 0, 25, 2, 10, 3000, False),
                                                                       (16, 31, 0, 15, 1, 10, 100, True)])
def test_rowsplit_native_parameter_corners(gpu, oracle, nd, block, mind, uniq, d12, tex, spk, outputs_pc):
    """Negative / positive minDisparity (the clamped border columns feed the L/R check), uniqueness and L/R check off, speckle filter off or
    with a threshold above most components (every tile then has border components that must be kept), the largest block, float disparity
    only: 3 tiles, the stitched frame against the oracle chain."""
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    w, h = 520, 231
    calib = synth.scaled_calibration(w, h)
    kw = _kw(nd, block)
    kw.update(min_disparity=mind, uniqueness_ratio=uniq, disp12_max_diff=d12, texture_threshold=tex)
    lraw, rraw = _frames(w, h, max(16, nd - 8), 61, "mono8", calib)
    outputs = capi.TOPIC_DISPARITY | (capi.TOPIC_POINTCLOUD if outputs_pc else 0)
    rs = RS.RowSplit(w, h, calib, capi.BMParams(**kw), 3, "mono8", outputs, spk, 1.5)
    rs.upload(lraw, rraw)
    rs.run()
    d16, f32, pc = rs.download()
    rs.close()
    e16, e32, epc = _oracle_chain(oracle, lraw, rraw, calib, w, h, kw, "mono8", spk, 24)
    assert np.array_equal(d16, e16) and np.array_equal(f32, e32)
    if outputs_pc:
        _check((d16, f32, pc), (e16, e32, epc))
    else:
        assert pc is None
    assert (e16 != (mind - 1) * 16).mean() > 0.05
