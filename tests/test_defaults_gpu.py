"""Round 5: the fast configuration without environment variables.

 * rows that are not whole 128-byte lines: embeddings the library allocates itself pad their stride to 128 bytes when that costs
   at most 8 % over the reference's 16-byte padding (reference embedding.cpp:43-50; csrc/embedding.cpp: align_embedding_dim).
   Nothing a user sees changes: shape, gather and training results (bit for bit against the oracle, which keeps the reference's
   stride), files — a table saved from a 128-byte-aligned embedding loads into a 16-byte-aligned one and back.
 * the placement probe of wholememory_malloc as an argument: create_embedding(..., placement_probe="auto") /
   wholememory_ext_set_malloc_probe."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype_name,dim,auto_stride,ref_stride", [
    ("float32", 602, 608, 604), ("float32", 300, 320, 300), ("float32", 513, 544, 516), ("float32", 200, 200, 200),
    ("float32", 100, 100, 100), ("float32", 127, 128, 128), ("float32", 128, 128, 128), ("float32", 1000, 1024, 1000),
    ("float16", 301, 320, 304), ("float16", 100, 104, 104), ("float32", 7, 8, 8)])
def test_row_stride_defaults(gpu_env, knobs, dtype_name, dim, auto_stride, ref_stride):
    import torch
    import wholegraph_amd.torch as wgth
    for setting, want in ((None, auto_stride), ("auto", auto_stride), ("16", ref_stride)):
        if setting is None:
            knobs.unset("WM_EMBEDDING_ROW_ALIGN")
        else:
            knobs.set("WM_EMBEDDING_ROW_ALIGN", setting)
        emb = wgth.create_embedding(gpu_env, "chunked", "cuda", getattr(torch, dtype_name), [1001, dim])
        assert emb.get_embedding_tensor().stride() == (want, 1) and emb.shape == (1001, dim), (setting, emb.get_embedding_tensor().stride())
        wgth.destroy_embedding(emb)


@pytest.mark.parametrize("mt", ["chunked", "distributed", "continuous"])
@pytest.mark.parametrize("kind,params", [("sgd", {"weight_decay": 0.01}), ("adam", {}), ("adagrad", {})])
def test_training_on_a_line_padded_table_keeps_the_reference_bits(gpu_env, knobs, tmp_path, mt, kind, params):
    """dim 300 (word2vec): 1200-byte rows, stride 320 floats by default. Two training steps against the oracle on the
    reference's stride (300), then save -> load into a table created with WM_EMBEDDING_ROW_ALIGN=16 -> save again: same bytes."""
    import torch
    import wholegraph_amd.torch as wgth
    knobs.unset("WM_EMBEDDING_ROW_ALIGN")
    n_rows, dim, n_idx = 30011, 300, 40003
    emb = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    assert emb.get_embedding_tensor().stride() == (320, 1)
    rng = np.random.default_rng(17)
    init = rng.standard_normal((n_rows, dim)).astype(np.float32)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.from_numpy(init).cuda())
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    module = wgth.WholeMemoryEmbeddingModule(emb)
    module.train()
    tab = oracle.ShardedTable.from_full(init.copy(), 1)
    tab.dim = dim
    ref_opt = oracle.Optimizer(kind, n_rows, dim, **params)
    for step in range(2):
        idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
        idx[::4] = idx[0]
        w = rng.standard_normal((n_idx, dim)).astype(np.float32)
        out = module(torch.from_numpy(idx).cuda())
        exp_out = np.zeros((n_idx, dim), np.float32)
        oracle.gather(tab, idx, exp_out)
        assert out.detach().cpu().numpy().tobytes() == exp_out.tobytes()
        (out * torch.from_numpy(w).cuda()).sum().backward()
        opt.step(0.02)
        oracle.gradient_apply(tab, [ref_opt], [idx], [w], 0.02)
        torch.cuda.synchronize()
        assert local.cpu().numpy().tobytes() == tab.shards[0][:, :dim].tobytes(), "step %d" % step
    prefix = str(tmp_path / "padded")
    emb.get_embedding_tensor().to_file_prefix(prefix)
    raw = np.fromfile(prefix + "_part_0_of_1", dtype=np.float32)
    assert raw.size == n_rows * dim and raw.tobytes() == tab.shards[0][:, :dim].tobytes()   # logical rows, no padding in the file
    knobs.set("WM_EMBEDDING_ROW_ALIGN", "16")
    emb16 = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    assert emb16.get_embedding_tensor().stride() == (300, 1)
    emb16.get_embedding_tensor().from_file_prefix(prefix)
    torch.cuda.synchronize()
    l16, _ = emb16.get_embedding_tensor().get_local_tensor()
    assert l16.cpu().numpy().tobytes() == raw.tobytes()
    prefix16 = str(tmp_path / "packed")
    emb16.get_embedding_tensor().to_file_prefix(prefix16)
    knobs.unset("WM_EMBEDDING_ROW_ALIGN")
    emb128 = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    emb128.get_embedding_tensor().from_file_prefix(prefix16)
    torch.cuda.synchronize()
    l128, _ = emb128.get_embedding_tensor().get_local_tensor()
    assert l128.stride() == (320, 1) and l128.cpu().numpy().tobytes() == raw.tobytes()
    wgth.destroy_wholememory_optimizer(opt)
    for e in (emb, emb16, emb128):
        wgth.destroy_embedding(e)


def test_placement_probe_as_an_argument(gpu_env, knobs):
    """create_embedding(..., placement_probe="auto") probes the shard whatever WM_MALLOC_PROBE says (unset here), the next
    creation without the argument does not, and the ext entry refuses what it does not understand."""
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    knobs.unset("WM_MALLOC_PROBE")
    rows = (5 << 28) // 512            # 1.25 GiB of 512-byte rows: above the probe's 1 GiB floor
    probed = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [rows, 128], placement_probe="auto")
    plain = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [rows, 128])
    was = lambda e: wmb.lib().wholememory_ext_handle_was_probed(e.get_embedding_tensor()._handle())
    assert was(probed) == 1 and was(plain) == 0
    idx = torch.randperm(rows, device="cuda")[:100000]        # distinct rows: a scatter of duplicates has no defined winner
    src = torch.randn(100000, 128, device="cuda")
    probed.get_embedding_tensor().scatter(src, idx)
    assert torch.equal(probed.gather(idx), src)
    assert wmb.lib().wholememory_ext_set_malloc_probe(b"sometimes") == wmb.lib().wholememory_ext_set_malloc_probe(b"9") != 0
    assert wmb.lib().wholememory_ext_set_malloc_probe(b"env") == 0
    for e in (probed, plain):
        wgth.destroy_embedding(e)


def test_unprobed_trained_table_is_mentioned_once(wm_lib):
    """an optimizer on a big device table that was allocated without the probe: ONE line naming the knob (per process)"""
    code = r'''
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows = (5 << 28) // 512
for probe in (None, None, "auto"):
    emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, 128], placement_probe=probe)
    opt = wgth.create_wholememory_optimizer(emb, "sgd", {})
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)
print("DONE")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    assert (r.stdout + r.stderr).count("allocated without the placement probe") == 1, r.stdout + r.stderr
