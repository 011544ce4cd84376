#!/usr/bin/env python3
"""State-machine walks beyond the collected ones (tests/test_gpu_state_machine.py::_walk): seeds 100..123, 200 steps each,
every configuration, with and without the pipelined step; VFM_CHECK_WREC on.  Prints one line per failing walk and a count.
usage: python tools/extended_walks.py [first_seed] [n_seeds] [n_steps]        (GPU box; ~3 minutes)"""
import os
import sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("VFM_CHECK_WREC", "1")
import test_gpu_state_machine as SM

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ok = bad = 0
for cfg in SM.CONFIGS:
    for seed in range(first, first + n):
        for pipe in ((False, True) if cfg == "F2_d32" else (False,)):      # (the pipelined step: the configuration the collected tests walk it on)
            try:
                SM._walk(cfg, seed, pipe, n_steps=steps)
                ok += 1
            except AssertionError as e:
                bad += 1
                print(f"FAIL cfg {cfg} seed {seed} pipeline {pipe}: {str(e)[:300]}", flush=True)
print(f"walks: {ok + bad} ({len(SM.CONFIGS)} configurations x {n} seeds in the bitwise forms + {n} seeds with the pipelined step on F2_d32, {steps} steps each); failed: {bad}")
