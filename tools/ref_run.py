"""Drive the imported reference EvalModel over a synthetic clip (fixture generation only)."""
import numpy as np
import torch

from tools.ref_import import build_reference_model
from otvm_amd.synth_weights import synthetic_state_dict
from otvm_amd.synth_data import synthetic_clip, soft_alpha


def frame_inputs(frames, t, trimap=None, alpha=None):
    fg = torch.from_numpy(frames[t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()  # DataLoader collate copies
    H, W = frames.shape[1:3]
    if alpha is None:
        a = torch.ones(1, 1, 1, H, W)
    else:
        a = torch.from_numpy(alpha)[None, None, None]
    tri_gt = None if trimap is None else torch.from_numpy(trimap)[None, None]
    return a, fg, fg.clone(), tri_gt


def run_reference(H, W, T, seed=0, skip=5, max_num=5, dilate_kernel=12, style="demo", wseed=0, hooks=None):
    m = build_reference_model(dilate_kernel)
    m.load_state_dict(synthetic_state_dict(wseed), strict=True)
    frames, tri = synthetic_clip(H, W, T, seed)
    outs = []
    for t in range(T):
        if style == "demo":
            a, fg, bg, tri_gt = frame_inputs(frames, t, trimap=tri)
        else:
            a, fg, bg, tri_gt = frame_inputs(frames, t, trimap=None, alpha=soft_alpha(H, W, t))
        memorize = (t % skip == 0) if skip > 2 else False
        out = m(a, fg, bg, tri=None, tri_gt=tri_gt, first_frame=(t == 0), last_frame=(t == T - 1),
                memorize=memorize, max_memory_num=max_num, large_input=False)
        bank = 0 if m.memories["key"] is None else m.memories["key"].shape[3]
        outs.append(dict(alpha=out[3][0, 0, 0].numpy().copy(), trimap=out[1][0, 0].numpy().copy(), bank=bank))
    return m, outs


if __name__ == "__main__":
    import sys, time
    H, W, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    t0 = time.time()
    m, outs = run_reference(H, W, T)
    print("time", time.time() - t0)
    for t, o in enumerate(outs):
        a = o["alpha"]; tr = o["trimap"]
        cls = tr.argmax(0)
        print(t, "bank", o["bank"], "alpha mean %.3f min %.3f max %.3f frac0 %.3f frac1 %.3f" % (a.mean(), a.min(), a.max(), (a == 0).mean(), (a == 1).mean()),
              "tri frac", [(cls == i).mean().round(3) for i in range(3)], "conf", np.sort(tr, 0)[-1].mean().round(3))
