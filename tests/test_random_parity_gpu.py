"""Randomised parity sweep on the GPU: many random (dtype pair, dim, stride, column offset, output stride, index
dtype, negative / duplicate ids, memory type) combinations of gather and scatter through the C ABI, each
compared bit-exactly with the oracle. Seeds are fixed; the sweep is repeated twice in one process to catch
state-dependent (flaky) behaviour."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

FLOATS = [np.float32, np.float16, np.float64]
INTS = [np.int8, np.int16, np.int32, np.int64]


def _tt(np_dtype):
    import torch
    return {np.float32: torch.float32, np.float16: torch.float16, np.float64: torch.float64, np.int8: torch.int8,
            np.int16: torch.int16, np.int32: torch.int32, np.int64: torch.int64}[np_dtype]


def one_case(comm, rng, case_id):
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
    fam = FLOATS if rng.random() < 0.6 else INTS
    tdt, pdt = fam[rng.integers(len(fam))], fam[rng.integers(len(fam))]
    idt = [np.int32, np.int64][rng.integers(2)]
    dim = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 127, 128, 129, 256, 300, 512, 513, 1024, 1030]))
    col_off = int(rng.choice([0, 0, 1, 3, 4, 8]))
    stride = dim + col_off + int(rng.choice([0, 0, 1, 4, 5]))
    n_rows = int(rng.integers(1, 5000))
    n_idx = int(rng.choice([0, 1, 63, 64, 65, 1000, 4097]))
    out_stride = dim + int(rng.choice([0, 0, 2, 3]))
    mt = ["continuous", "chunked", "distributed"][rng.integers(3)]
    if mt == "distributed" and (col_off + dim > stride):
        col_off = 0
    root = wgth.create_wholememory_tensor(comm, mt, "cuda", [n_rows, stride], _tt(tdt), [stride, 1])
    full = (rng.integers(-100, 100, (n_rows, stride))).astype(tdt)
    local, _ = root.get_local_tensor()
    local.copy_(torch.from_numpy(full).cuda())
    view = root.get_sub_tensor([0, col_off], [n_rows, col_off + dim]) if (col_off or stride != dim) else root
    tab = oracle.ShardedTable([full.copy()], np.array([0, n_rows], dtype=np.uint64), dim, stride, col_off)
    idx = rng.integers(0, n_rows, n_idx).astype(idt)
    if n_idx > 3:
        idx[rng.integers(0, n_idx, max(1, n_idx // 10))] = -1
        idx[1] = idx[2]
    out_np = rng.integers(-3, 3, (max(n_idx, 1), out_stride)).astype(pdt)[:n_idx]
    out_t = torch.from_numpy(out_np.copy()).cuda()
    out_view = out_t[:, :dim] if out_stride != dim else out_t
    wi, wo = wrap_torch_tensor(torch.from_numpy(idx).cuda()), wrap_torch_tensor(out_view)
    wmb.check(wmb.lib().wholememory_gather(view.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                           C.c_void_p(get_stream()), -1))
    torch.cuda.synchronize()
    exp = out_np.copy()
    oracle.gather(tab, idx, exp, dim=dim, out_stride=out_stride)
    assert out_t.cpu().numpy().tobytes() == exp.tobytes(), \
        "gather case %d: %s->%s dim=%d stride=%d off=%d n=%d os=%d %s %s" % (
            case_id, tdt.__name__, pdt.__name__, dim, stride, col_off, n_idx, out_stride, idt.__name__, mt)
    # scatter unique ids (duplicates would race) from the same plain tensor back into the table
    if n_idx > 0:
        uniq = np.unique(idx[idx >= 0])
        sidx = idx.copy()
        seen = set()
        for i, v in enumerate(sidx):
            if v >= 0:
                if int(v) in seen:
                    sidx[i] = -1
                seen.add(int(v))
        wi2 = wrap_torch_tensor(torch.from_numpy(sidx).cuda())
        wmb.check(wmb.lib().wholememory_scatter(wo.handle, wi2.handle, view.wmb_tensor, get_wholegraph_env_fns(),
                                                C.c_void_p(get_stream()), -1))
        torch.cuda.synchronize()
        oracle.scatter(exp, sidx, tab, dim=dim, in_stride=out_stride)
        assert local.cpu().numpy().tobytes() == tab.shards[0].tobytes(), "scatter case %d" % case_id
        del uniq
    if view is not root:
        wgth.destroy_wholememory_tensor(view)
    wgth.destroy_wholememory_tensor(root)


@pytest.mark.parametrize("seed", [0, 1])
def test_random_gather_scatter_sweep(gpu_env, seed):
    rng = np.random.default_rng(1000 + seed)
    for rep in range(2):
        for case_id in range(150):
            one_case(gpu_env, rng, rep * 1000 + case_id)
