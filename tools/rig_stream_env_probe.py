"""The one-call rig tracker in a process that already holds other HIP streams (argument: none | torch | streams): does the
tracker's second stream still run beside the first?  (tracker.hip create_side_stream: candidates are tried.)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import torch
mode = sys.argv[1]
keep = []
if mode == "torch":
    x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
if mode == "streams":
    x = torch.zeros(1, device="cuda")
    keep = [torch.cuda.Stream() for _ in range(8)]
    for s in keep:
        with torch.cuda.stream(s): y = x + 1
    torch.cuda.synchronize()
import run_rig_tracker
sys.argv = ["x", "radtan", "2", "1200", "20"]
run_rig_tracker.main()
