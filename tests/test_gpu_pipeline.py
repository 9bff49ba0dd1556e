"""GPU test (-m gpu): the device-resident batch pipeline (gip_batch_run) equals the oracle chain on every frame."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_chain(O, lr, rr, calib, w, h, po, spk_size, spk_diff):
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    return ml, mr


@pytest.mark.parametrize("encoding", ["mono8", "bayer_rggb8", "bgr8"])
def test_batch_pipeline_matches_oracle(gpu, oracle, encoding):
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    O = oracle
    w, h, nd, B = 416, 160, 64, 3
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=9, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=60 + i, calib=calib)[:2] for i in range(B)]
    if encoding == "mono8":
        L = np.stack([r[0] for r in raws]); R = np.stack([r[1] for r in raws])
    elif encoding == "bgr8":
        L = np.stack([synth.colorize(r[0], 1) for r in raws]); R = np.stack([synth.colorize(r[1], 2) for r in raws])
    else:
        L = np.stack([synth.bayer_mosaic(synth.colorize(r[0], 1), encoding) for r in raws])
        R = np.stack([synth.bayer_mosaic(synth.colorize(r[1], 2), encoding) for r in raws])
    pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, encoding, capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 100, 2.0)
    pipe.install_maps(pipe.host_maps_to_device())
    pipe.run(torch.from_numpy(L).to(gpu), torch.from_numpy(R).to(gpu))
    torch.cuda.synchronize()
    bm_ms, total_ms, runs = pipe.timing_read()
    assert runs == 1 and 0 < bm_ms <= total_ms
    d16 = pipe.disparity.cpu().numpy(); d32 = pipe.disparity_f32.cpu().numpy(); pc = pipe.pointcloud.cpu().numpy()
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    Q = O.stereo_q(calib["left"]["P"], calib["right"]["P"])
    for b in range(B):
        ml_mono = O.convert_color(L[b], encoding, "mono8")
        mr_mono = O.convert_color(R[b], encoding, "mono8")
        col = O.convert_color(L[b], encoding, "bgr8")
        rl, rr = O.remap(ml_mono, *ml), O.remap(mr_mono, *mr)
        exp = O.filter_speckles(O.stereobm(rl, rr, O.bm_params(**kw)), -16, 100, 32)
        assert np.array_equal(d16[b], exp), (encoding, b)
        e32 = O.disparity_to_f32(exp, calib["left"]["P"][2], calib["right"]["P"][2])
        assert np.array_equal(d32[b], e32)
        epc = O.pack_pointcloud2(O.reproject_to_3d(e32, Q, True), O.remap(col, *ml))
        gx, ex = pc[b][..., :12].copy().view(np.float32), epc[..., :12].copy().view(np.float32)
        assert np.array_equal(np.isnan(gx), np.isnan(ex))
        ok = ~np.isnan(ex)
        assert np.allclose(gx[ok], ex[ok], rtol=2.4e-7, atol=0)  # float xyz: <= 1 ulp (direct vs accumulated column term)
        assert np.array_equal(pc[b][..., 12:], epc[..., 12:])
        assert (exp != -16).mean() > 0.2
    pipe.close()


def test_batch_pipeline_odd_size_batch_of_four(gpu, oracle):
    """Odd width and height (scalar fallbacks of the speckle scan, unaligned plane rows) with a batch of four (the remap form that
    shares the table decode between four images)."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    O = oracle
    w, h, nd, B = 333, 131, 48, 4
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=11, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=700 + i, calib=calib)[:2] for i in range(B)]
    L = np.stack([r[0] for r in raws]); R = np.stack([r[1] for r in raws])
    pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 60, 2.0)
    pipe.install_maps(pipe.host_maps_to_device())
    pipe.run(torch.from_numpy(L).to(gpu), torch.from_numpy(R).to(gpu))
    torch.cuda.synchronize()
    d16 = pipe.disparity.cpu().numpy(); d32 = pipe.disparity_f32.cpu().numpy(); pc = pipe.pointcloud.cpu().numpy()
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    Q = O.stereo_q(calib["left"]["P"], calib["right"]["P"])
    for b in range(B):
        rl, rr = O.remap(L[b], *ml), O.remap(R[b], *mr)
        exp = O.filter_speckles(O.stereobm(rl, rr, O.bm_params(**kw)), -16, 60, 32)
        assert np.array_equal(d16[b], exp), b
        e32 = O.disparity_to_f32(exp, calib["left"]["P"][2], calib["right"]["P"][2])
        assert np.array_equal(d32[b], e32)
        epc = O.pack_pointcloud2(O.reproject_to_3d(e32, Q, True), O.gray_to_bgr(rl))
        gx, ex = pc[b][..., :12].copy().view(np.float32), epc[..., :12].copy().view(np.float32)
        assert np.array_equal(np.isnan(gx), np.isnan(ex))
        ok = ~np.isnan(ex)
        assert np.allclose(gx[ok], ex[ok], rtol=2.4e-7, atol=0)
        assert np.array_equal(pc[b][..., 12:], epc[..., 12:])
        assert (exp != -16).mean() > 0.1
    pipe.close()


@pytest.mark.parametrize("block", [9, 15, 21])
def test_pipelines_in_flight_with_shared_matcher_stream(gpu, block):
    """bench.py keeps three pipelines in flight (own streams, own buffers) whose matcher kernels go to ONE shared stream
    (gip_batch_desc::matcher_stream).  Every pipeline must still produce exactly what it produces alone.  At block 15 / 21 the two runs
    also use the two forms of the matcher's row staging (alone: refetch, with a matcher stream: ring; k_stereobm_c4_impl.h launch_c4)."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    w, h, nd, B = 480, 160, 64, 3
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    pipes = [BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 200, 4.0) for _ in range(3)]
    ins = []
    for k, p in enumerate(pipes):
        p.install_maps(p.host_maps_to_device())
        raws = [synth.raw_pair(w, h, nd, seed=900 + 10 * k + i, calib=calib) for i in range(B)]
        ins.append((torch.from_numpy(np.stack([r[0] for r in raws])).to(gpu), torch.from_numpy(np.stack([r[1] for r in raws])).to(gpu)))
    # reference: each pipeline alone, synchronised
    alone = []
    for p, (l, r) in zip(pipes, ins):
        p.run(l, r)
        torch.cuda.synchronize()
        alone.append((p.disparity.clone(), p.disparity_f32.clone(), p.pointcloud.clone()))
        p.disparity.zero_(); p.disparity_f32.zero_(); p.pointcloud.zero_()
    streams = [torch.cuda.Stream(device=gpu) for _ in pipes]
    shared = torch.cuda.Stream(device=gpu)
    torch.cuda.synchronize()
    for rep in range(4):                       # several rounds back to back, nothing synchronised in between
        for k, (p, (l, r)) in enumerate(zip(pipes, ins)):
            with torch.cuda.stream(streams[k]):
                p.run(l, r, shared)
    torch.cuda.synchronize()
    for k, p in enumerate(pipes):
        assert torch.equal(p.disparity, alone[k][0]), k
        assert torch.equal(p.disparity_f32, alone[k][1]), k
        assert torch.equal(p.pointcloud.view(torch.int32), alone[k][2].view(torch.int32)), k
        assert (p.disparity != -16).float().mean() > 0.2
    for p in pipes:
        p.close()


def test_concurrent_stream_set_and_front_back_phases(gpu):
    """gip_concurrent_streams_create hands out streams that were measured to dispatch side by side (at least the copy-in / compute / copy-out
    trio of the streamer; four with the runtime's default four hardware queues), and gip_batch_desc::phase splits the chain at the matcher:
    FRONT(k+1) enqueued before BACK(k) on ONE stream, the matchers on a second one -- the result must be what every pipeline produces alone.
    Also the misuse cases: BACK without FRONT, FRONT twice, bad arguments."""
    import ctypes as C
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    lib = capi.lib()
    arr = (C.c_void_p * 4)()
    assert lib.gip_concurrent_streams_create(0, 0, arr, None) == 1 and lib.gip_concurrent_streams_create(0, 17, arr, None) == 1
    assert lib.gip_concurrent_streams_create(0, 2, None, None) == 1 and lib.gip_concurrent_streams_create(99, 2, arr, None) == 5
    idle = [torch.cuda.Stream(device=gpu) for _ in range(3)]      # streams created before must not matter
    handles, independent = capi.concurrent_streams(4, 0)
    assert len(set(handles)) == 4 and all(handles) and 2 <= independent <= 4      # (4 with the default four hardware queues on an idle device)
    handles6, independent6 = capi.concurrent_streams(6, 0)        # more than the hardware queues: still six usable streams
    assert len(set(handles6)) == 6 and 1 <= independent6 <= 6
    capi.release_streams(handles6, 0)
    st, shared = torch.cuda.ExternalStream(handles[0], device=gpu), torch.cuda.ExternalStream(handles[1], device=gpu)
    w, h, nd, B = 480, 160, 64, 3
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=9, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    pipes = [BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 200, 4.0) for _ in range(2)]
    ins = []
    for k, p in enumerate(pipes):
        p.install_maps(p.host_maps_to_device())
        raws = [synth.raw_pair(w, h, nd, seed=700 + 10 * k + i, calib=calib) for i in range(B)]
        ins.append((torch.from_numpy(np.stack([r[0] for r in raws])).to(gpu), torch.from_numpy(np.stack([r[1] for r in raws])).to(gpu)))
    alone = []
    for p, (l, r) in zip(pipes, ins):
        p.run(l, r)
        torch.cuda.synchronize()
        alone.append((p.disparity.clone(), p.disparity_f32.clone(), p.pointcloud.clone()))
        p.disparity.zero_(); p.disparity_f32.zero_(); p.pointcloud.zero_()
    torch.cuda.synchronize()
    with pytest.raises(capi.GipError):
        pipes[0].run(*ins[0], shared, phase=capi.BATCH_BACK)                  # nothing to complete
    n = 7
    with torch.cuda.stream(st):
        pipes[0].run(*ins[0], shared, phase=capi.BATCH_FRONT)
        with pytest.raises(capi.GipError):
            pipes[0].run(*ins[0], shared, phase=capi.BATCH_FRONT)             # the previous FRONT is still open
        with pytest.raises(capi.GipError):
            pipes[0].run(*ins[0], shared)                                     # ... also for a whole run
        for i in range(n):
            if i + 1 < n:
                pipes[(i + 1) % 2].run(*ins[(i + 1) % 2], shared, phase=capi.BATCH_FRONT)
            pipes[i % 2].run(*ins[i % 2], shared, phase=capi.BATCH_BACK)
    torch.cuda.synchronize()
    for k, p in enumerate(pipes):
        assert torch.equal(p.disparity, alone[k][0]), k
        assert torch.equal(p.disparity_f32, alone[k][1]), k
        assert torch.equal(p.pointcloud.view(torch.int32), alone[k][2].view(torch.int32)), k
    # FRONT phases in order on one stream, BACK phases on another, matchers on the third: the handle orders its own buffers by events
    for p in pipes:
        p.disparity.zero_(); p.disparity_f32.zero_(); p.pointcloud.zero_()
    torch.cuda.synchronize()
    sf, sb, sm3 = [torch.cuda.ExternalStream(hd, device=gpu) for hd in handles[:3]]
    for i in range(9):
        with torch.cuda.stream(sf):
            pipes[i % 2].run(*ins[i % 2], sm3, phase=capi.BATCH_FRONT)
        with torch.cuda.stream(sb):
            pipes[i % 2].run(*ins[i % 2], sm3, phase=capi.BATCH_BACK)
    torch.cuda.synchronize()
    for k, p in enumerate(pipes):
        assert torch.equal(p.disparity, alone[k][0]), k
        assert torch.equal(p.disparity_f32, alone[k][1]), k
        assert torch.equal(p.pointcloud.view(torch.int32), alone[k][2].view(torch.int32)), k
    del sf, sb, sm3
    # phases on the caller's own stream (no matcher stream): FRONT then BACK is the whole chain
    pipes[0].disparity.zero_()
    pipes[0].run(*ins[0], phase=capi.BATCH_FRONT)
    pipes[0].run(*ins[0], phase=capi.BATCH_BACK)
    torch.cuda.synchronize()
    assert torch.equal(pipes[0].disparity, alone[0][0])
    for p in pipes:
        p.close()
    del st, shared, idle
    capi.release_streams(handles, 0)


@pytest.mark.parametrize("encoding", ["bayer_rggb8", "bgr8", "rgb8"])
def test_front_back_phases_colour_encodings(gpu, encoding):
    """FRONT / BACK on colour inputs: the colour conversion, the rectified colour plane of the cloud and the fused front end all sit in the FRONT half;
    BACK must find them.  Compared with the whole run of the same pipeline."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    w, h, nd, B = 320, 128, 48, 2
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=11, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=40 + i, calib=calib)[:2] for i in range(B)]
    if encoding.startswith("bayer"):
        L = np.stack([synth.bayer_mosaic(synth.colorize(r[0], 1), encoding) for r in raws])
        R = np.stack([synth.bayer_mosaic(synth.colorize(r[1], 2), encoding) for r in raws])
    else:
        L = np.stack([synth.colorize(r[0], 1) for r in raws])
        R = np.stack([synth.colorize(r[1], 2) for r in raws])
    l, r = torch.from_numpy(np.ascontiguousarray(L)).to(gpu), torch.from_numpy(np.ascontiguousarray(R)).to(gpu)
    for fused in (False, True):
        pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, encoding, capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 100, 4.0, fused_front_end=fused)
        pipe.install_maps(pipe.host_maps_to_device())
        pipe.run(l, r)
        torch.cuda.synchronize()
        whole = (pipe.disparity.clone(), pipe.disparity_f32.clone(), pipe.pointcloud.clone())
        pipe.disparity.zero_(); pipe.disparity_f32.zero_(); pipe.pointcloud.zero_()
        shared = torch.cuda.Stream(device=gpu)
        pipe.run(l, r, shared, phase=capi.BATCH_FRONT)
        pipe.run(l, r, shared, phase=capi.BATCH_BACK)
        torch.cuda.synchronize()
        assert torch.equal(pipe.disparity, whole[0]) and torch.equal(pipe.disparity_f32, whole[1])
        assert torch.equal(pipe.pointcloud.view(torch.int32), whole[2].view(torch.int32))
        assert (pipe.disparity != -16).float().mean() > 0.2
        pipe.close()


def test_config3_1080p_bayer_nd256_block21_speckle(gpu, oracle):
    """BASELINE.json configs[2]: 1920x1080 bayer_rggb8 pair, ndisp 256, block 21 + speckle filter, full chain on one GPU.
    One frame is checked bit-exactly against the oracle (a few seconds of CPU)."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    O = oracle
    w, h, nd, B = 1920, 1080, 256, 2
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=21, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=80 + i, calib=calib)[:2] for i in range(B)]
    L = np.stack([synth.bayer_mosaic(synth.colorize(r[0], 1), "bayer_rggb8") for r in raws])
    R = np.stack([synth.bayer_mosaic(synth.colorize(r[1], 2), "bayer_rggb8") for r in raws])
    pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "bayer_rggb8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 800, 5.0)
    pipe.install_maps(pipe.host_maps_to_device())
    pipe.run(torch.from_numpy(L).to(gpu), torch.from_numpy(R).to(gpu))
    torch.cuda.synchronize()
    d16 = pipe.disparity.cpu().numpy()
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    b = 1
    rl = O.remap(O.bayer_to_gray(L[b], "bayer_rggb8"), *ml)
    rr = O.remap(O.bayer_to_gray(R[b], "bayer_rggb8"), *mr)
    exp = O.filter_speckles(O.stereobm(rl, rr, O.bm_params(**kw)), -16, 800, 80)
    assert np.array_equal(d16[b], exp)
    assert (exp != -16).mean() > 0.2
    pc = pipe.pointcloud.cpu().numpy()[b]
    assert np.array_equal(pc[..., 16:19], O.remap(O.bayer_to_bgr(L[b], "bayer_rggb8"), *ml))
    assert np.array_equal(np.isnan(pc[..., :4].copy().view(np.float32)[..., 0]) | (exp == 0), (exp == -16) | (exp == 0))
    pipe.close()


def test_config5_4k_nd192_block15_properties(gpu, oracle):
    """BASELINE.json configs[4] size (4096x2160 mono8, ndisp 192, block 15) on ONE GPU (the 4-GPU row-tile split is not
    built yet): a horizontal band is checked bit-exactly against the oracle run on the same band (StereoBM rows are
    independent beyond the block radius), plus range / ROI invariants on the full frame."""
    from ros_gpu_stereo_processor_amd import kernels as K
    import torch
    w, h, nd, block = 4096, 2160, 192, 15
    l, r, _ = synth.rectified_pair(w, h, nd, seed=91, nrect=8)
    kw = dict(block_size=block, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    got = K.stereobm(torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu), K.bm_params(**kw)).cpu().numpy()
    valid = got != -16
    assert got[valid].min() >= 0 and got[valid].max() <= (nd - 1) * 16 and valid.mean() > 0.3
    roi = oracle.valid_disparity_roi(w, h, oracle.bm_params(**kw))
    mask = np.ones((h, w), bool)
    mask[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]] = False
    assert (got[mask] == -16).all()
    # band of 200 output rows: the oracle needs the band plus (block radius + 1 prefilter row) of context on each side
    y0, y1, ctx = 900, 1100, block // 2 + 2
    band = oracle.stereobm(l[y0 - ctx:y1 + ctx], r[y0 - ctx:y1 + ctx], oracle.bm_params(**kw))
    assert np.array_equal(got[y0:y1], band[ctx:ctx + (y1 - y0)])


@pytest.mark.parametrize("shape,nd,block,world", [((540, 1024), 64, 9, 4), ((2160, 4096), 192, 15, 4), ((301, 640), 32, 21, 3)])
def test_rowsplit_equals_full_frame(gpu, shape, nd, block, world):
    """BASELINE.json configs[4] decomposition: the frame split into row tiles with halos (what each rank holds after the halo
    exchange), matched tile by tile and stitched, equals the full-frame result bit for bit."""
    import torch
    from ros_gpu_stereo_processor_amd import kernels as K, rowsplit as RS
    h, w = shape
    l, r, _ = synth.rectified_pair(w, h, nd, seed=95, nrect=6)
    lt, rt = torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu)
    p = K.bm_params(block_size=block, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    full = K.stereobm(lt, rt, p)
    split = RS.split_stereobm_simulated(lt, rt, p, world, block)
    assert torch.equal(full, split)
    assert (full != -16).float().mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("prefilter_type", [1, 0])
def test_batch_pipeline_fused_front_end_equals_two_pass(gpu, prefilter_type):
    """gip_batch_run with the one-pass rectify + XSOBEL front end (gip_batch_desc::fused_front_end) and with the default two passes give the
    same disparity, float disparity and cloud bytes (the cloud's colour comes from the rectified plane the fused pass writes as a
    by-product); with the NORMALIZED_RESPONSE prefilter both runs take the two-pass form."""
    import torch
    from ros_gpu_stereo_processor_amd import capi, synth
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    w, h, nd, B = 500, 131, 48, 4
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=prefilter_type, prefilter_size=9, prefilter_cap=31, block_size=11, min_disparity=0, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=90 + i, calib=calib)[:2] for i in range(B)]
    L = torch.from_numpy(np.stack([r[0] for r in raws])).to(gpu)
    R = torch.from_numpy(np.stack([r[1] for r in raws])).to(gpu)
    outs = []
    for fused in (True, False):
        pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 60, 2.0, fused_front_end=fused)
        pipe.install_maps(pipe.host_maps_to_device())
        pipe.run(L, R)
        torch.cuda.synchronize()
        outs.append((pipe.disparity.cpu().numpy().copy(), pipe.disparity_f32.cpu().numpy().copy(), pipe.pointcloud.cpu().numpy().copy()))
        pipe.close()
    assert (outs[0][0] != -16).mean() > 0.2
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


@pytest.mark.gpu
def test_rowsplit_speckle_simulated_ranks(gpu, oracle):
    """configs[4] speckle stage on the HIP labeler: a frame cut into 2, 3 and 5 row tiles, components merged across the tile borders
    from the exported border labels (what the ranks all-gather), equals the filter on the whole frame -- on a real disparity map and
    on the synthetic frame whose speckles straddle the borders."""
    import torch
    from ros_gpu_stereo_processor_amd import kernels as K, rowsplit as RS
    w, h, nd = 420, 150, 48
    l, r, _ = synth.rectified_pair(w, h, nd, seed=77)
    po = oracle.bm_params(block_size=9, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    d = oracle.stereobm(l, r, po)
    frames = [(d, 60, 32)]
    rng = np.random.default_rng(4)
    a = np.full((42, 40), -16, np.int16)
    a[2:40, 2:38] = (rng.integers(0, 2, (38, 36)) * 640 + 160).astype(np.int16)
    a[12:17, 5:8] = 3000
    a[6:36, 20] = 5000
    a[26:28, 30:34] = 7000
    a[28:30, 30:32] = 7016
    frames.append((a, 16, 32))
    for frame, max_size, max_diff in frames:
        exp = oracle.filter_speckles(frame.copy(), -16, max_size, max_diff)
        whole = K.filter_speckles(torch.from_numpy(frame.copy()).to(gpu), -16, max_size, max_diff)
        assert np.array_equal(whole.cpu().numpy(), exp)
        for world in (2, 3, 5):
            got = RS.split_speckle_simulated(torch.from_numpy(frame.copy()).to(gpu), world, -16, max_size, max_diff)
            assert np.array_equal(got.cpu().numpy(), exp), world


@pytest.mark.gpu
def test_rowsplit_frame_single_rank_equals_pipeline(gpu, oracle):
    """rowsplit.rowsplit_frame (the per-rank step of `bench.py --rowsplit`) with world size 1: rectify, match, speckle, cloud of the whole
    frame through the same code path the ranks run -- disparity equal to the oracle chain, cloud NaN mask consistent."""
    import torch
    from ros_gpu_stereo_processor_amd import kernels as K, rowsplit as RS, capi
    from ros_gpu_stereo_processor_amd import dist as D
    w, h, nd = 480, 130, 48
    calib = synth.scaled_calibration(w, h)
    lr, rr, _, _ = synth.raw_pair(w, h, nd, seed=9, calib=calib)
    maps = []
    for side in ("left", "right"):
        m1, m2 = D.build_maps_host(calib[side], w, h)
        maps += [torch.from_numpy(m1).to(gpu), torch.from_numpy(m2.view(np.int16)).to(gpu)]
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=11, min_disparity=0, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15,
              disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    Q = oracle.stereo_q(calib["left"]["P"], calib["right"]["P"])
    y0, y1, d, f32, cloud = RS.rowsplit_frame(torch.from_numpy(lr).to(gpu), torch.from_numpy(rr).to(gpu), maps, capi.BMParams(**kw), 0, 1, 60, 32, Q,
                                              calib["left"]["P"][2], calib["right"]["P"][2])
    ml = oracle.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], w, h)
    mr = oracle.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], w, h)
    exp = oracle.filter_speckles(oracle.stereobm(oracle.remap(lr, *ml), oracle.remap(rr, *mr), oracle.bm_params(**kw)), -16, 60, 32)
    assert (y0, y1) == (0, h) and np.array_equal(d.cpu().numpy(), exp) and (exp != -16).mean() > 0.2
    assert np.array_equal(f32.cpu().numpy(), oracle.disparity_to_f32(exp, calib["left"]["P"][2], calib["right"]["P"][2]))
    nan_mask = np.isnan(cloud.cpu().numpy()[..., :4].copy().view(np.float32))[..., 0]
    assert np.array_equal(nan_mask, exp == -16)


def _run_selftest(nproc, port, rehearse):
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if rehearse:
        env["GIP_BENCH_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "tools", "nccl_selftest.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "selftest ok" in out.stdout


@pytest.mark.gpu
def test_multirank_paths_three_processes_one_gpu(gpu):
    """The N > 1 code paths with REAL processes: model broadcast of the frame-sharded mode, and the row-tile split of one frame (halo
    exchange by batched isend/irecv, distributed speckle with the HIP labeler, border all-gather) over three ranks -- all on device 0 with
    gloo as the transport, because RCCL refuses two ranks on one device.  The stitched disparity must equal the single-GPU frame."""
    _run_selftest(3, 29518, rehearse=True)


@pytest.mark.gpu
def test_nccl_broadcast_and_halo_exchange_two_devices(gpu):
    """The same over RCCL ("nccl" backend).  Needs two visible devices; the 1-GPU test box skips it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's 8-GPU node runs the N > 1 bench)")
    _run_selftest(2, 29517, rehearse=False)


@pytest.mark.gpu
@pytest.mark.parametrize("nd,mind", [(64, 0), (48, 5)])
def test_cloud_table_is_bit_identical_to_the_arithmetic(gpu, nd, mind):
    """The batch pipeline's cloud kernel reads 1 / (Q33 + Q32 d), Z and the float disparity from a table indexed by the disparity value
    (gip_w_table_build, k_post.hip); the stateless launcher gipk_disparity_outputs computes them per pixel.  Same expressions, so the same
    bytes: float disparity and every one of the 32 bytes of every PointCloud2 record."""
    import torch
    from ros_gpu_stereo_processor_amd import capi, kernels as K
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    w, h, B = 416, 160, 2
    calib = synth.scaled_calibration(w, h)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=9, min_disparity=mind, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    raws = [synth.raw_pair(w, h, nd, seed=70 + i, calib=calib)[:2] for i in range(B)]
    L = torch.from_numpy(np.stack([r[0] for r in raws])).to(gpu)
    R = torch.from_numpy(np.stack([r[1] for r in raws])).to(gpu)
    pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 100, 2.0)
    maps = pipe.host_maps_to_device()
    pipe.install_maps(maps)
    pipe.run(L, R)
    torch.cuda.synchronize()
    m = pipe.proc.model()
    rect_l = K.remap(L, maps[0], maps[1])
    for b in range(B):
        f32, pc = K.disparity_outputs(pipe.disparity[b].contiguous(), rect_l[b].contiguous(), np.array(m["Q"], np.float64).reshape(4, 4), m["cx_l"], m["cx_r"], mind)
        assert torch.equal(f32, pipe.disparity_f32[b])
        assert torch.equal(pc, pipe.pointcloud[b])
        assert (pipe.disparity[b] != (mind - 1) * 16).float().mean() > 0.2
    pipe.close()
