import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, "/root/repo")
import torch, bench
sys.stdout = sys.stderr
r = bench.arnn_extra()
print(r)
