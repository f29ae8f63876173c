"""Multi-GPU inference, rehearsed on one GPU: the tile list is sharded over W "ranks" that run one after the other, the
all-reduce of the exchange step is replaced by the sum of the ranks' packed buffers, and the assembled label map must be
the single-process annonet_infer() result (planes to float summation-order tolerance; labels equal except exact near-ties)."""
import numpy as np
import pytest

import annonet_amd as aa
from annonet_amd import dist as aad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_inference_assembles_the_single_process_result(world):
    import torch
    dev = torch.device("cuda:0")
    t = aa.TrainingNet(1, 3, aa.ANH_FP32, seed=4)
    t.SetNetWidth(0.5, 4); t.SetClassCount(3); t.Initialize()
    net = t.GetRuntimeNet(aa.ANH_FP32)
    stream = aad.handle_stream(net)   # torch work below is enqueued on the net's own stream
    rng = np.random.default_rng(world)
    H, W = 230, 301
    ov = t.GetRequiredInputDimension()
    tp = aa.tiling.parameters(96, 112, ov, ov)
    tiles = aa.tiling.get_tiles(W, H, tp)
    assert len(tiles) >= 2 * world
    image = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).to(dev)
    gains = [0.0, 0.01, -0.02]

    torch.cuda.synchronize()
    with torch.cuda.stream(stream):   # one stream for the net's kernels and the torch kernels of the rehearsal
        want_lab = torch.zeros((H, W), dtype=torch.int16, device=dev)
        want_pl = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
        aa.annonet_infer_device(net, image.data_ptr(), H, W, want_lab.data_ptr(), want_pl.data_ptr(), gains=gains, tiling_parameters=tp)

        ex = aad.OverlapExchange(tiles, world, W, H, dev)
        assert 0 < ex.pixels() < H * W // 2
        planes, packed = [], []
        for r in range(world):       # step 1 on every "rank": blend its own tiles (no labels yet)
            pl = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
            aa.annonet_infer_device(net, image.data_ptr(), H, W, 0, pl.data_ptr(), gains=gains, tiling_parameters=tp, tiles=aad.shard_tiles(tiles, r, world))
            planes.append(pl); packed.append(ex.pack(pl))
        total = packed[0].clone()
        for p in packed[1:]:
            total += p                # what dist.all_reduce(SUM) leaves on every rank
        owner = aad.tile_owner(len(tiles), world)
        got_lab = np.full((H, W), -1, dtype=np.int64)
        got_pl = np.zeros((3, H, W), dtype=np.float32)
        bands, gather0 = [], None    # step 4: LabelGather — the coded row bands rank 0 receives, merged there
        host_maps = []               # step 4': HostLabelMap — every rank copies the cells of its own tiles into ONE shared host map
        for r in range(world):       # steps 2b + 3: scatter the sums back, label the rank's rows
            ex.unpack(planes[r], total)
            lab = torch.zeros((H, W), dtype=torch.int16, device=dev)
            mine = aad.shard_tiles(tiles, r, world)
            row0, row1 = max(0, min(x[0][1] for x in mine)), min(H, max(x[0][3] for x in mine) + 1)
            aa.argmax_device(net, planes[r].data_ptr(), H, W, row0, row1, lab.data_ptr(), gains=gains)
            g = aad.LabelGather(tiles, world, r, W, H, dev, 3)
            gather0 = gather0 or g
            bands.append(g.code(lab))
            host_maps.append(aad.HostLabelMap(tiles, world, r, W, H, name=host_maps[0].name if host_maps else None))
            host_maps[-1].deliver(lab.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            lab_np, pl_np = lab.cpu().numpy().view(np.uint16), planes[r].cpu().numpy()
            for i, (full, _) in enumerate(tiles):   # a rank answers for the pixels its tiles cover
                if owner[i] != r:
                    continue
                l, tp_, rr, b = max(full[0], 0), max(full[1], 0), min(full[2], W - 1), min(full[3], H - 1)
                sub = got_lab[tp_:b + 1, l:rr + 1]
                new = lab_np[tp_:b + 1, l:rr + 1].astype(np.int64)
                assert ((sub == -1) | (sub == new)).all()      # ranks that share a pixel agree on its label
                got_lab[tp_:b + 1, l:rr + 1] = new
                got_pl[:, tp_:b + 1, l:rr + 1] = pl_np[:, tp_:b + 1, l:rr + 1]
    assert (got_lab >= 0).all()
    np.testing.assert_array_equal(aad.LabelGather.decode(gather0.merge(bands)).cpu().numpy().view(np.uint16).astype(np.int64), got_lab)   # the ONE map of the job
    # ... and the ONE host map the ranks filled themselves: the same labels, every byte written by exactly one rank
    np.testing.assert_array_equal(host_maps[0].array.astype(np.int64), got_lab)
    assert sum(m.bytes for m in host_maps) == H * W * 2 and all(m.pinned for m in host_maps)
    for m in reversed(host_maps):
        m.close()
    want_pl_np, want_lab_np = want_pl.cpu().numpy(), want_lab.cpu().numpy().view(np.uint16).astype(np.int64)
    span = float(want_pl_np.max() - want_pl_np.min())
    np.testing.assert_allclose(got_pl, want_pl_np, rtol=0, atol=2e-6 * span)      # (a+b)+(c+d) vs ((a+b)+c)+d in four-tile corners
    diff = got_lab != want_lab_np
    if diff.any():
        g = np.asarray(gains, dtype=np.float32)[:, None, None]
        srt = np.sort(want_pl_np + g, axis=0)
        assert ((srt[-1] - srt[-2])[diff] <= 4e-6 * span).all()
    assert diff.mean() < 1e-4
    # ... and against the ORACLE (annonet_infer.cpp:42-214 restated on the CPU), not only against the library's own single-process result
    from oracle.oracle import OracleNet
    o = OracleNet(1, 3, 3, 0.5, 4)
    p, r = t.get_params()
    o.params[:], o.running[:] = p, r
    ref_lab, ref_pl = o.infer(image.cpu().numpy(), gains=gains, max_tile=(96, 112), overlap=ov, want_blended=True)
    np.testing.assert_allclose(got_pl, ref_pl, rtol=0, atol=2e-6 * span)
    odiff = got_lab != ref_lab.astype(np.int64)
    if odiff.any():
        srt = np.sort(ref_pl + np.asarray(gains, dtype=np.float32)[:, None, None], axis=0)
        assert ((srt[-1] - srt[-2])[odiff] <= 4e-6 * span).all()
    assert odiff.mean() < 1e-4


def test_exchange_is_empty_for_one_rank():
    import torch
    tiles = aa.tiling.get_tiles(300, 200, aa.tiling.parameters(96, 112, 19, 19))
    ex = aad.OverlapExchange(tiles, 1, 300, 200, torch.device("cuda:0"))
    assert ex.pixels() == 0
    ex.run(torch.zeros((3, 200, 300), device="cuda:0"))     # no process group needed


def test_the_default_stream_cannot_be_named_by_set_stream():
    """torch's default stream has the handle 0, which set_stream reads as "a stream of your own": ordering torch work with a
    handle goes through handle_stream() instead (the data-parallel all-reduce and the overlap exchange rely on it)."""
    import torch
    t = aa.TrainingNet(1, 3, aa.ANH_FP32, seed=1)
    t.SetNetWidth(0.25, 4); t.SetClassCount(2); t.Initialize()
    own = t.stream_ptr()
    assert own != 0
    t.set_stream(torch.cuda.current_stream().cuda_stream)        # 0 on the default stream
    assert t.stream_ptr() != 0
    side = torch.cuda.Stream()
    t.set_stream(side.cuda_stream)
    assert t.stream_ptr() == side.cuda_stream
    assert aad.handle_stream(t).cuda_stream == side.cuda_stream
    t.set_stream(None)
    assert t.stream_ptr() not in (0, side.cuda_stream)


def head_forms_agree(epilogue, kernel, what="head forms"):
    """bf16 planes and labels of the two places the 1x1 head can run: the same bf16 activations and fp32-equivalent weights, another
    summation order (MFMA accumulation against a chain of fp32 multiply-adds)."""
    pa, pb = epilogue["bf16_blended"], kernel["bf16_blended"]
    span = float(pb.max() - pb.min())
    assert span > 0
    assert float(np.abs(pa - pb).max()) <= 2e-5 * span, (what, float(np.abs(pa - pb).max()), span)
    for key in ("bf16_labels", "bf16_labels_streamed"):
        assert float((epilogue[key] != kernel[key]).mean()) <= 1e-3, (what, key)
    np.testing.assert_array_equal(epilogue["bf16_labels_streamed"], epilogue["bf16_labels"])


def test_tile_batch_size_does_not_change_the_result(tmp_path):
    """annonet_infer() runs tiles of equal size through the net as batches (ANH_INFER_TILE_BATCH; default: equal batches of at most 16) and blends them one after the
    other in list order: label map and blended planes are BIT-identical to the tile-by-tile loop of annonet_infer.cpp:116-164, in both
    precisions, and the streamed host form (strips up, label rows down, batches within a tile row) gives the same label map."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    runs = {}
    for batch in ("1", "4", "3"):
        out = str(tmp_path / f"b{batch}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out], env=dict(os.environ, ANH_INFER_TILE_BATCH=batch),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[batch] = np.load(out)
    assert int(runs["1"]["bf16_tiles"]) >= 9
    for batch in ("4", "3"):
        for key in ("bf16_labels", "bf16_blended", "fp32_labels", "fp32_blended", "bf16_labels_streamed", "fp32_labels_streamed"):
            np.testing.assert_array_equal(runs[batch][key], runs["1"][key], err_msg=f"batch {batch}: {key}")
    for prec in ("bf16", "fp32"):
        np.testing.assert_array_equal(runs["4"][prec + "_labels_streamed"], runs["4"][prec + "_labels"])
    # the 1x1 head in the epilogue of the last hidden layer's conv (default; two MFMAs on a two-term bf16 split of the fp32 head weights)
    # against the separate fused head/blend kernel (ANH_HEAD_IN_EPILOGUE=0; fp32 multiply-adds chained over the channels): the same
    # activations and weights summed in another order — planes within 2e-5 of their span, labels equal but for near-ties; the fp32 mode
    # has one form
    out = str(tmp_path / "head_kernel.npz")
    r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out], env=dict(os.environ, ANH_HEAD_IN_EPILOGUE="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sep = np.load(out)
    for key in ("fp32_labels", "fp32_blended"):
        np.testing.assert_array_equal(sep[key], runs["4"][key], err_msg=f"head kernel: {key}")
    head_forms_agree(sep, runs["4"])
    # the XCD-band tile walk against the grid-stride walk, and the 32-channel conv's filter fragments in registers against LDS reads
    # every item: the same values to the same addresses
    # ... and one blend launch per tile against one per tile batch
    for env in ({"ANH_WS_XCD_BANDS": "0"}, {"ANH_WS_FILTER_REGS": "0"}, {"ANH_BLEND_BATCH": "0"}):
        out = str(tmp_path / ("variant_" + "_".join(env) + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out, "1.0", "1", "3", "2"], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        variant = np.load(out)
        if "ref_w1" not in runs:
            out0 = str(tmp_path / "ref_w1.npz")
            r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out0, "1.0", "1", "3", "2"], env=dict(os.environ),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            runs["ref_w1"] = np.load(out0)
        for key in ("bf16_labels", "bf16_blended", "bf16_labels_streamed"):
            np.testing.assert_array_equal(variant[key], runs["ref_w1"][key], err_msg=f"{env}: {key}")
    # ... on the nets whose last hidden layer has the 32 channels that form covers (width 1.0), for every class count it takes
    for classes in ("1", "2", "3", "4"):
        pair = {}
        for name, env in (("epilogue", {}), ("kernel", {"ANH_HEAD_IN_EPILOGUE": "0"})):
            out = str(tmp_path / f"w1_{classes}_{name}.npz")
            r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out, "1.0", "1", classes, "2"], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            pair[name] = np.load(out)
        head_forms_agree(pair["epilogue"], pair["kernel"], f"{classes} classes")
        assert np.isfinite(pair["epilogue"]["bf16_blended"]).all() and float(np.abs(pair["epilogue"]["bf16_blended"]).max()) > 0
    # the other bf16 inference form (raw conv outputs stored, consumers re-apply bn + relu: ANH_INFER_POST_ACT=0) differs by bf16
    # rounding points only; the fp32 mode does not have two forms
    out = str(tmp_path / "raw.npz")
    r = subprocess.run([sys.executable, os.path.join(here, "helpers", "run_tiled_infer.py"), out], env=dict(os.environ, ANH_INFER_POST_ACT="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = np.load(out)
    np.testing.assert_array_equal(raw["fp32_labels"], runs["4"]["fp32_labels"])
    np.testing.assert_array_equal(raw["fp32_blended"], runs["4"]["fp32_blended"])
    span = float(runs["4"]["bf16_blended"].max() - runs["4"]["bf16_blended"].min())
    assert np.abs(raw["bf16_blended"] - runs["4"]["bf16_blended"]).max() <= 0.02 * span
    assert (raw["bf16_labels"] != runs["4"]["bf16_labels"]).mean() <= 0.03     # random-init net: near-ties everywhere
