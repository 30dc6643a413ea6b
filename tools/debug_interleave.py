import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(1, os.path.join(ROOT, "code"))
import pytest
import test_module_state_gpu as T

class MP:
    def setenv(self, k, v): os.environ[k] = v

for graph in (False,):
    a, _ = T.run_steps(4, False, MP(), graph)
    b, _ = T.run_steps(4, False, MP(), graph)
    c, _ = T.run_steps(4, True, MP(), graph)
    for s in range(4):
        sa, sb, sc = np.array(a[s][1]), np.array(b[s][1]), np.array(c[s][1])
        print(graph, s, "plain vs plain", np.abs(sa - sb).max(), "plain vs interleaved", np.abs(sa - sc).max(), "gen", T.rel(c[s][0], a[s][0]))
        print("   ", np.round(sa, 5)); print("   ", np.round(sc, 5))
