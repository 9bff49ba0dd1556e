"""The streaming C-ABI (gip_streamer_*): pinned ring, three HIP streams, host-callback delivery.  Streamed outputs must equal what
the synchronous per-frame schedule (gip_process_pair = StereoProcessor::imageCb, reference src/StereoProcessor.cpp:157-298) leaves in
the slots for the same pair, in submission order, for every ring depth / batch size incl. a partial last batch."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


def _processor(w, h, nd, block):
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    calib = synth.scaled_calibration(w, h)
    proc = P.GpuStereoProcessor(0)
    proc.initStereoModel(P.camera_info(w, h, **calib["left"]), P.camera_info(w, h, **calib["right"]))
    proc.setParams(capi.BMParams(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=nd,
                                 texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0))
    proc.setMaxSpeckleSize(60)
    proc.setMaxSpeckleDiff(2.0)
    return proc, calib


@pytest.mark.parametrize("depth,fpb,nframes,adaptive,lanes", [(3, 1, 7, False, 1), (2, 2, 5, False, 2), (4, 3, 10, False, 0), (0, 0, 12, True, 0), (3, 4, 11, True, 3), (2, 2, 9, True, 1)])
def test_streamed_outputs_equal_process_pair(gpu, depth, fpb, nframes, adaptive, lanes):
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    w, h, nd = 400, 120, 48
    proc, calib = _processor(w, h, nd, 9)
    frames = [synth.raw_pair(w, h, nd, seed=200 + i, calib=calib)[:2] for i in range(nframes)]
    # reference results: the synchronous schedule, one pair at a time
    exp = []
    for l, r in frames:
        proc.processPair(l, r, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD)
        d16 = proc.downloadMat(P.L_DISPARITY)
        f32 = proc.downloadMat(P.L_DISPARITY_32F)
        sid = proc.enqueueSendPoints(P.L_POINTS2, P.L_RECT_COLOR)
        proc.waitForAllStreams()
        exp.append((d16.copy(), f32.copy(), proc.pointCloudMessage(sid)["data"].copy()))
        proc.cleanSenders()
    assert (exp[0][0] != -16).mean() > 0.2
    got = {}
    order = []

    def on_result(seq, header, d16, f32, cloud):
        order.append(seq)
        got[seq] = (header.seq, d16, f32, cloud)
    st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, want_disparity16=True, depth=depth, frames_per_batch=fpb, on_result=on_result, adaptive=adaptive, compute_lanes=lanes)
    for i, (l, r) in enumerate(frames):
        hdr = capi.Header(seq=1000 + i, stamp_sec=i, stamp_nsec=7, frame_id=b"cam")
        st.submit(l, r, hdr)
    st.flush()
    assert st.delivered() == nframes
    assert 1 <= st.launches() <= nframes and (adaptive or st.launches() == -(-nframes // max(fpb, 1)))
    assert order == list(range(nframes))                       # delivery in submission order
    for i in range(nframes):
        hseq, d16, f32, cloud = got[i]
        assert hseq == 1000 + i
        assert np.array_equal(d16, exp[i][0]), i
        assert np.array_equal(f32, exp[i][1]), i
        assert np.array_equal(cloud.view(np.uint8), exp[i][2].view(np.uint8)), i
    # the ring keeps working after a flush
    st.submit(*frames[0])
    st.flush()
    assert st.delivered() == nframes + 1 and np.array_equal(got[nframes][1], exp[0][0])
    st.close()
    proc.close()


def test_streamer_counts_without_callback_and_rejects_bad_config(gpu):
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    w, h, nd = 320, 96, 32
    proc, calib = _processor(w, h, nd, 7)
    l, r = synth.raw_pair(w, h, nd, seed=3, calib=calib)[:2]
    st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY, depth=3, frames_per_batch=4, adaptive=False)
    for _ in range(9):
        st.submit(l, r)
    st.flush()
    assert st.delivered() == 9
    st.close()
    # pinned inputs are read in place by the H2D copy (no staging memcpy): same results
    import torch
    pl, pr = torch.from_numpy(l).pin_memory(), torch.from_numpy(r).pin_memory()
    got = []
    ref = []
    for pinned, sink, a, b in ((False, ref, l, r), (True, got, pl.numpy(), pr.numpy())):
        st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY, depth=2, frames_per_batch=2, inputs_pinned=pinned, on_result=lambda seq, hdr, d16, f32, pc, sink=sink: sink.append(f32))
        for _ in range(5):
            st.submit(a, b)
        st.flush()
        st.close()
    assert len(got) == 5 and all(np.array_equal(x, ref[0]) for x in got + ref)
    with pytest.raises(capi.GipError):
        Streamer(proc, w + 2, h, "mono8", capi.TOPIC_DISPARITY)      # size differs from the calibration
    with pytest.raises(capi.GipError):
        Streamer(proc, w, h, "mono8", 0)                              # nothing to deliver
    proc.close()


def test_streamed_bayer_colour_cloud_equals_process_pair(gpu):
    """Bayer input: debayer to mono (both eyes) and to BGR (left, the cloud's colour) inside the streamed chain."""
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    w, h, nd = 384, 128, 32
    proc, calib = _processor(w, h, nd, 11)
    enc = "bayer_grbg8"
    frames = []
    for i in range(5):
        l, r = synth.raw_pair(w, h, nd, seed=400 + i, calib=calib)[:2]
        frames.append((synth.bayer_mosaic(synth.colorize(l, 1), enc), synth.bayer_mosaic(synth.colorize(r, 2), enc)))
    exp = []
    for l, r in frames:
        proc.processPair(l, r, enc, capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD)
        f32 = proc.downloadMat(P.L_DISPARITY_32F)
        sid = proc.enqueueSendPoints(P.L_POINTS2, P.L_RECT_COLOR)
        proc.waitForAllStreams()
        exp.append((f32.copy(), proc.pointCloudMessage(sid)["data"].copy()))
        proc.cleanSenders()
    got = {}
    st = Streamer(proc, w, h, enc, capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, depth=3, frames_per_batch=2, on_result=lambda seq, hdr, d16, f32, pc: got.__setitem__(seq, (f32, pc)))
    for l, r in frames:
        st.submit(l, r)
    st.flush()
    st.close()
    assert sorted(got) == list(range(5))
    for i in range(5):
        assert np.array_equal(got[i][0], exp[i][0]), i
        assert np.array_equal(got[i][1].view(np.uint8), exp[i][1].view(np.uint8)), i
    assert len(np.unique(exp[0][1][..., 16:19].reshape(-1, 3), axis=0)) > 50        # the cloud really carries colour
    proc.close()


def test_streamer_failed_launch_is_sticky_and_shapes_are_checked(gpu):
    """A launch that gip_batch_run rejects (the block size is made larger than the image after open) must not leave a half-filled slot behind:
    the streamer drains, reports the error from submit and flush from then on, and still closes cleanly (no write past a slot, no hang).
    Images of the wrong shape never reach the C side."""
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    w, h, nd = 320, 96, 32
    proc, calib = _processor(w, h, nd, 7)
    l, r = synth.raw_pair(w, h, nd, seed=3, calib=calib)[:2]
    for adaptive, fpb in ((True, 0), (False, 2)):
        proc.setBlockSize(7)
        got = []
        st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY, depth=2, frames_per_batch=fpb, adaptive=adaptive, compute_lanes=1, on_result=lambda *a: got.append(a[0]))
        st.submit(l, r)
        st.submit(l, r)
        st.flush()
        assert st.delivered() == 2
        with pytest.raises(ValueError):
            st.submit(l[:-1], r[:-1])
        with pytest.raises(ValueError):
            st.submit(np.stack([l] * 3, -1), np.stack([r] * 3, -1))
        proc.setBlockSize(255)                  # valid for the setter, impossible for a 96-row image: StereoBM validation fails inside the launch
        with pytest.raises(capi.GipError):
            for _ in range(3):                  # fixed batching launches on the second pair
                st.submit(l, r)
        for _ in range(3):                      # sticky: the same error, nothing written anywhere
            with pytest.raises(capi.GipError):
                st.submit(l, r)
        with pytest.raises(capi.GipError):
            st.flush()
        assert st.delivered() == 2
        st.close()
    proc.setBlockSize(7)
    st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY)      # a fresh streamer on the same processor works again
    st.submit(l, r)
    st.flush()
    assert st.delivered() == 1
    st.close()
    proc.close()


def test_streamer_camera_rate_gets_batches_of_one_and_a_burst_gets_full_batches(gpu):
    """Adaptive batching: a caller that submits slowly (a camera) finds the compute lane free and the copy-out idle every time -- every pair is
    launched by itself and delivered without waiting for a batch to fill or for a flush; a caller that submits as fast as it can gets batches that
    fill up (far fewer launches than pairs)."""
    import time
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    w, h, nd = 640, 240, 64
    proc, calib = _processor(w, h, nd, 7)
    l, r = synth.raw_pair(w, h, nd, seed=5, calib=calib)[:2]
    st = Streamer(proc, w, h, "mono8", capi.TOPIC_DISPARITY, frames_per_batch=16)
    for i in range(12):
        st.submit(l, r)
        deadline = time.time() + 2.0
        while st.delivered() < i + 1 and time.time() < deadline:      # delivered WITHOUT flush: nothing waits for a batch to fill
            time.sleep(0.0005)
        assert st.delivered() == i + 1, "pair %d was not delivered on its own" % i
        time.sleep(0.003)
    assert st.launches() == 12
    # a burst: the lane / the copy-out are busy most of the time, the batches fill
    n = 16 * 20
    l0 = st.launches()
    for _ in range(n):
        st.submit(l, r)
    st.flush()
    assert st.delivered() == 12 + n
    assert st.launches() - l0 <= n // 2, (st.launches() - l0, n)
    st.close()
    proc.close()
