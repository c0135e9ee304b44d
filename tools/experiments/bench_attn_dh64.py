"""Head size 64 (a width-512, 8-head model): attention forward / backward launch times on a few shapes (not a test)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench_attn import case
for shape in ((64, 256, 8, 64), (64, 257, 8, 64), (64, 128, 8, 64), (32, 512, 8, 64), (64, 256, 8, 32)):
    case(*shape, 30)
case(64, 256, 8, 64, 30, q_limit=1)
