"""Probe: the MS-G3D step (forward + backward) recorded into HIP graphs, several per process -- do the replays reproduce the eager
loss?     python tools/probes/msg3d_graph_probe.py f32,bf16x3 [keeploss,again,empty,fwd]

Measured (profiles/README.md, round 2): every capture replays bit-identically to the eager step UNLESS the loss tensor of an
earlier eager step is still referenced at capture time (flag ``keeploss``).  That reference keeps the parameters' gradient
accumulator nodes alive, pinned to the stream they were created on; the second and later captures in the process then replay
garbage that changes from replay to replay (the first capture happens to survive).  Rule taken from it: drop every reference to
an earlier step's autograd graph before capturing, and check the replayed loss against the eager one (bench.py and
tools/msg3d_bench.py do both)."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from fusion_gcn_amd import ops
from fusion_gcn_amd.datasets.utd_mhad import constants as utd
from fusion_gcn_amd.models.msg3d.msg3d import Model
from fusion_gcn_amd.util import Graph
flags = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else set()
dev = torch.device("cuda:0")
shape, classes = (1, 128, 20, 3), 27
torch.manual_seed(1)
model = Model({"skeleton": shape}, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint)).to(dev).train()
x = torch.randn(8, *shape, device=dev); y = torch.randint(0, classes, (8,), device=dev)
params = list(model.parameters())
work = torch.cuda.Stream(); work.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(work)
keep = []
def eager_steps(n):
    ls = []
    for _ in range(n):
        for p in params: p.grad = None
        loss = F.cross_entropy(model(x), y); loss.backward(); ls.append(round(float(loss), 5))
    return ls
for mode in sys.argv[1].split(","):
    with ops.math_mode(mode):
        if "keeploss" in flags:
            for _ in range(3):
                for p in params: p.grad = None
                loss = F.cross_entropy(model(x), y); loss.backward()
            print(mode, "eager", float(loss))
        else:
            print(mode, "eager", eager_steps(3), flush=True)
        for p in params: p.grad = None
        if "empty" in flags:
            torch.cuda.synchronize(); torch.cuda.empty_cache()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if "fwd" in flags:
                with torch.no_grad():
                    gl = F.cross_entropy(model(x), y)
            else:
                gl = F.cross_entropy(model(x), y); gl.backward()
        keep.append((graph, gl))
        ls = []
        for _ in range(6):
            graph.replay(); ls.append(round(float(gl), 5))
        print(mode, "graph", ls, flush=True)
        if "again" in flags:
            print(mode, "eager again", eager_steps(2), flush=True)
