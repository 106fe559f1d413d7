import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.gae_sweep import time_gae
import json
for T, N in [(128, 1 << 18), (128, 1 << 20), (128, 1 << 21), (64, 1 << 20)]:
  print(json.dumps(time_gae(T, N)), flush=True)
