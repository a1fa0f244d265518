#!/usr/bin/env python
"""Randomised shapes for gather / scatter against torch indexing on the GPU (torch is the checker here: exact copies and
the same round-to-nearest casts): dtype pairs, dims 1..700, padded strides, column offsets (sub-tensor views), output
strides, int32 / int64 ids with negatives and duplicates, chunked / continuous / distributed (one rank).
usage: fuzz_rows.py [cases] [seed]      (FUZZ_WIDE=1: most cases draw any dim in 1..1400 or a few wider ones, up to 2561)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream

torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
comm = wgth.create_group_communicator(1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WIDE = os.environ.get("FUZZ_WIDE") == "1"
FLOATS = [torch.float32, torch.float16, torch.float64, torch.bfloat16]
INTS = [torch.int8, torch.int16, torch.int32, torch.int64]
bad = 0
for case in range(cases):
    fam = FLOATS if rng.random() < 0.6 else INTS
    tdt, odt = fam[rng.integers(len(fam))], fam[rng.integers(len(fam))]
    if torch.bfloat16 in (tdt, odt) and tdt != odt:
        tdt = odt = torch.bfloat16 if rng.random() < 0.5 else torch.float32   # bf16 is not registered for casts (as in the reference)
    dim = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 127, 128, 129, 200, 256, 300, 513, 602, 700]))
    if WIDE and rng.random() < 0.7:   # any width up to rows of 5.6 KiB (both chunk sizes of the LDS-staged kernels and past them)
        dim = int(rng.integers(1, 1400)) if rng.random() < 0.8 else int(rng.choice([1024, 1030, 1280, 1281, 1279, 2048, 2560, 2561]))
    stride = dim + int(rng.choice([0, 0, 1, 3, 4, 13]))
    col0 = int(rng.integers(0, stride - dim + 1))
    n_rows = int(rng.integers(1, 30000))
    n = int(rng.choice([0, 1, 5, 63, 64, 65, 1000, 4097, 20000]))
    idt = torch.int32 if rng.random() < 0.5 else torch.int64
    mt = ["chunked", "continuous", "distributed"][rng.integers(3)]
    ostride = dim + int(rng.choice([0, 0, 2, 5]))
    desc = "case %d: %s table %s -> %s, rows %d dim %d stride %d col0 %d, n %d %s, out stride %d" % (
        case, mt, tdt, odt, n_rows, dim, stride, col0, n, idt, ostride)
    if os.environ.get('FUZZ_VERBOSE'):
        print(desc, flush=True)
    try:
        root = wgth.create_wholememory_tensor(comm, mt, "cuda", [n_rows, stride], tdt, [stride, 1])
        local, _ = root.get_local_tensor()
        if tdt.is_floating_point:
            local.copy_((torch.randn(n_rows, stride, device="cuda") * 100).to(tdt))
        else:
            info = torch.iinfo(tdt)
            local.copy_(torch.randint(max(info.min, -2 ** 31), min(info.max, 2 ** 31 - 1), (n_rows, stride), device="cuda").to(tdt))
        view = root.get_sub_tensor([0, col0], [n_rows, col0 + dim]) if (col0 or dim != stride) else root
        idx = torch.from_numpy(rng.integers(0, n_rows, n)).to(idt).cuda()
        if n > 4:
            idx[::7] = -1
            idx[1] = idx[3]
        outbuf = torch.full((max(n, 1), ostride), 7, dtype=odt, device="cuda")
        out = outbuf[:n, :dim]
        wi = wrap_torch_tensor(idx)
        # wrap_torch_tensor takes shape/stride from the tensor: a [n, dim] view with row stride `ostride`
        wo = wrap_torch_tensor(out)
        wmb.check(wmb.lib().wholememory_gather(view.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
        torch.cuda.synchronize()
        want = torch.full((n, dim), 7, dtype=odt, device="cuda")
        ok_rows = idx >= 0
        src = local[:, col0:col0 + dim]
        want[ok_rows] = src[idx[ok_rows].long()].to(odt)
        if not torch.equal(out.contiguous().view(torch.uint8), want.view(torch.uint8)) or not bool((outbuf[:, dim:] == 7).all()):
            bad += 1
            print("GATHER MISMATCH", desc, flush=True)
        # scatter the gathered rows (+1 where that is exact) back into distinct rows and compare the table
        if n > 0:
            ns = min(n, n_rows)
            perm = torch.randperm(n_rows, device="cuda")[:ns].to(idt)
            perm[::5] = -1
            before = local.clone()
            src_rows = outbuf[:ns, :dim]
            ws = wrap_torch_tensor(src_rows)
            wp = wrap_torch_tensor(perm)   # (the wrapper owns the C handle: it must outlive the call)
            wmb.check(wmb.lib().wholememory_scatter(ws.handle, wp.handle, view.wmb_tensor,
                                                    get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
            torch.cuda.synchronize()
            exp = before.clone()
            okp = perm >= 0
            exp[perm[okp].long(), col0:col0 + dim] = src_rows[okp].to(tdt)
            if not torch.equal(local.view(torch.uint8), exp.view(torch.uint8)):
                bad += 1
                print("SCATTER MISMATCH", desc, flush=True)
        if view is not root:
            wgth.destroy_wholememory_tensor(view)
        wgth.destroy_wholememory_tensor(root)
    except Exception as ex:   # noqa
        bad += 1
        print("ERROR", desc, repr(ex)[:300], flush=True)
print("cases %d, failures %d" % (cases, bad))
