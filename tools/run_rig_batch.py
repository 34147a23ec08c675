import sys, json
sys.path.insert(0, "/root/repo")
import bench
print(json.dumps(bench.rig_frontend_batch(), indent=None)[:2500])
