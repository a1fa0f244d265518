# round 6: the lane's self-test instead of the tool's environment variables. A gradient apply of 200 k ids under rocprofv3's
# counter collection (one kernel at a time) with the variables the library used to look for REMOVED from the process: the
# self-test has to notice (one WARN line) and the calls have to finish at once instead of sitting in a waiter until its timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/wst.py <<'PY'
import os, sys, time
for k in list(os.environ):
    if k.startswith("ROCPROF_COUNTER"): os.environ.pop(k)
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [2000000, 64])
wgth.create_wholememory_optimizer(emb, "sgd", {})
idx = torch.randint(0, 2000000, (200000,), device="cuda"); g = torch.randn((200000, 64), device="cuda")
t0 = time.time()
for _ in range(5):
    emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
torch.cuda.synchronize()
print("five gradient steps: %.2f s" % (time.time() - t0))
PY
echo "== plain process"; python /tmp/wst.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== under rocprofv3 --pmc FETCH_SIZE, ROCPROF_COUNTER* removed from the process"
rm -rf /tmp/wst; timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/wst -- python /tmp/wst.py 2>&1 | grep -v amdgpu.ids | grep -E "WARN|ERROR|five gradient|Traceback|Error" | head -8
