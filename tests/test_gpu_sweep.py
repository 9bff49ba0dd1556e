"""GPU test (-m gpu): EVERY (numDisparities, block size) cell of the column-sum matcher against the CPU oracle, bit for bit -- ndisp 16..256 in
steps of 16 x block 5..31 (224 shapes), widths that are not multiples of anything, batches of 9 (the two-images-per-wave kernels start at 8) and of
1, three min_disparity values and both settings of the L/R check in rotation.  The other matcher tests hold 32 fixed and 40 fuzzed cases; this sweep
is for changes that touch every instantiation (the row staging, the finalisation)."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nd", list(range(16, 257, 16)))
def test_every_block_size_of_this_disparity_count(gpu, oracle, nd):
    import torch
    from ros_gpu_stereo_processor_amd import kernels as K
    O = oracle
    for block in range(5, 32, 2):
        mind = (0, -8, 5)[(nd // 16 + block) % 3]
        w = nd + 3 * 50 + 37 + (block % 7)
        h = 2 * block + 41
        B = 9 if (nd + block) % 4 else 1
        kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=mind, num_disparities=nd, texture_threshold=10,
                  uniqueness_ratio=15, disp12_max_diff=1 if block % 4 != 1 else -1, speckle_window_size=0, speckle_range=0)
        pairs = [synth.rectified_pair(w, h, nd, seed=1000 + 7 * nd + block + i)[:2] for i in range(min(B, 2))]
        l = np.stack([pairs[i % len(pairs)][0] for i in range(B)])
        r = np.stack([pairs[i % len(pairs)][1] for i in range(B)])
        got = K.stereobm(torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu), K.bm_params(**kw)).cpu().numpy()
        po = O.bm_params(**kw)
        for i in range(min(B, 2)):
            exp = O.stereobm(l[i], r[i], po)
            assert np.array_equal(got[i], exp), (nd, block, mind, B, i, int((got[i] != exp).sum()))
        if B > 2:   # the rest of the batch repeats the two pairs: the last image (the odd one of the last pair of images) included
            assert np.array_equal(got[B - 1], got[(B - 1) % 2]) and np.array_equal(got[4], got[0]), (nd, block, "batch repeat")
        assert (got[0] != (mind - 1) * 16).mean() > 0.02, (nd, block, "nothing valid")


@pytest.mark.parametrize("block", [5, 9, 13, 17, 21, 25, 29, 31])
def test_both_row_staging_forms_agree_for_every_disparity_count(gpu, block):
    """The matcher keeps its rows in an LDS ring (launched on a matcher stream: the throughput pipelines) or fetches the leaving row again (on the
    caller's own stream); the sweep above checks the second form against the oracle, this one checks that the whole chain gives the same bytes in
    both -- every disparity count, batch 8 (the two-images-per-wave kernels included)."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    shared = torch.cuda.Stream(device=gpu)
    for nd in range(16, 257, 16):
        w, h, B = nd + 230, 2 * block + 50, 8
        calib = synth.scaled_calibration(w, h)
        kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=nd, texture_threshold=10,
                  uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
        pipe = BatchPipeline(w, h, calib, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY, 0, 0, 0.0, want_f32=False)
        pipe.install_maps(pipe.host_maps_to_device())
        raws = [synth.raw_pair(w, h, nd, seed=5000 + nd + block + i, calib=calib)[:2] for i in range(2)]
        l = torch.from_numpy(np.stack([raws[i % 2][0] for i in range(B)])).to(gpu)
        r = torch.from_numpy(np.stack([raws[i % 2][1] for i in range(B)])).to(gpu)
        pipe.run(l, r)
        torch.cuda.synchronize()
        own = pipe.disparity.clone()
        pipe.disparity.zero_()
        pipe.run(l, r, shared)
        torch.cuda.synchronize()
        assert torch.equal(pipe.disparity, own), (nd, block)
        assert (own != -16).float().mean() > 0.02, (nd, block)
        pipe.close()


@pytest.mark.parametrize("w,nd,mind", [(1537, 64, 0), (1700, 256, -8), (2049, 128, 5), (3000, 48, 0), (4096, 192, 0)])
def test_lr_check_with_one_dword_keys_on_wide_rows(gpu, oracle, w, nd, mind):
    """Rows wider than 1536 pixels run the L/R check (cv::validateDisparity) with 32-bit keys cost << 12 | (d - 16 minD) instead of the 64-bit
    (cost, x, d) keys (k_bm_finalize_r, k_post.hip): same winners, same ties.  Whole matcher + check against the oracle, batches of 1 and 8."""
    import torch
    from ros_gpu_stereo_processor_amd import kernels as K
    O = oracle
    h, block = 45, 11
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=mind, num_disparities=nd, texture_threshold=10,
              uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    pairs = [synth.rectified_pair(w, h, nd, seed=9000 + w + i)[:2] for i in range(2)]
    for B in (1, 8):
        l = np.stack([pairs[i % 2][0] for i in range(B)])
        r = np.stack([pairs[i % 2][1] for i in range(B)])
        got = K.stereobm(torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu), K.bm_params(**kw)).cpu().numpy()
        for i in range(min(B, 2)):
            exp = O.stereobm(l[i], r[i], O.bm_params(**kw))
            assert np.array_equal(got[i], exp), (w, nd, mind, B, i, int((got[i] != exp).sum()))
            # the check must have something to do: the same map without it differs
        kw2 = dict(kw, disp12_max_diff=-1)
        no_lr = K.stereobm(torch.from_numpy(l[:1]).to(gpu), torch.from_numpy(r[:1]).to(gpu), K.bm_params(**kw2)).cpu().numpy()
        assert (no_lr[0] != got[0]).sum() > 0, "the L/R check rejected nothing: the test does not exercise it"
