import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from cryo_ralib_amd import dist as rdist
rank, local, world = rdist.init_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for w in sys.argv[1:]:
    sub = bench.parse_args(["--workload", w, "--no-cpu-baseline", "--no-pcie", "--no-parity"])
    t0 = time.perf_counter()
    o = bench.run_workload(sub, 0, 0, 1, dev)
    print(w, round(o["value"]), "p/s", round(o["ms_per_step"], 2), "ms/step hot", round(o["roofline"]["hot_kernels_share_of_step"], 3), "wall", round(time.perf_counter() - t0, 1))
