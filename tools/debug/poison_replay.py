"""Debug aid: run tests/test_pack_replay_gpu.py's scenario with the caching allocator pre-filled with garbage, so that any
kernel that reads scratch it never wrote shows up as a difference (fresh processes hand out zero pages)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
val = float(os.environ.get("POISON", "nan"))
blocks = [torch.full((1 << 28,), val, device="cuda") for _ in range(24)]       # 24 GiB of garbage
small = [torch.full((n,), val, device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(8)]
del blocks, small
torch.cuda.synchronize()
import pytest
sys.exit(pytest.main(["-x", "-q", "tests/test_pack_replay_gpu.py"] + sys.argv[1:]))
