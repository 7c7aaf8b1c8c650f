import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
eng = model._get_engine()
plan = eng.plan(1080, 1920, 1)
for key, S in plan.steps.items():
    for st in S:
        lab = st[2] if isinstance(st[2], str) else str(st[-1])
        if "gn_apply" in lab or "gn_stats" in lab:
            print(key, lab)
