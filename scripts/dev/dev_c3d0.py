"""C3 at insert depth 0 alone (bench.py's instrumented leg): python scripts/dev/dev_c3d0.py [label ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
only = set(sys.argv[1:]) or {"C3_rgbd2mm_depth0"}
print(json.dumps(bench.other_configs(0, True, only=only), indent=1))
