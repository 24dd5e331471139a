"""The competition schedule shared by both layer families: which of the `flip_steps` post-warm-up steps run the competition
policy in one layer, drawn so that no step has more than `cap` competing layers.

One host-side restatement of `create_balanced_flip_current` (moe_model/model/moe/competesmoe.py:86-130;
moe_pretrain_model/layers/moe/competesmoe.py:161-205): one `torch.rand(1)` per slot on the SAME device RNG stream the
reference uses (cuda if available, else cpu) -- so a fixed `torch.manual_seed` reproduces the reference's `prob_flips` -- and a
slot that is already at the cap hands its competition step to the nearest free slot on the left, then on the right."""
from typing import Dict, List

import torch


def draw_balanced_flips(flip_steps: int, rate_flip: float, cap: int, previous: Dict[int, torch.Tensor]) -> List[bool]:
    rng_dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    freq = [0] * flip_steps
    for v in previous.values():
        for i, b in enumerate(v.tolist()):
            freq[i] += int(b)
    cur = [False] * flip_steps
    for i in range(flip_steps):
        if torch.rand(1, device=rng_dev).item() < rate_flip:
            if freq[i] < cap:
                cur[i] = True
                freq[i] += 1
                continue
            for j in list(range(i - 1, -1, -1)) + list(range(i + 1, flip_steps)):     # nearest free slot: left first, then right
                if freq[j] < cap and not cur[j]:
                    cur[j] = True
                    freq[j] += 1
                    break
    return cur
