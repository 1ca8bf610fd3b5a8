"""CPU ORACLE (test infrastructure): the conditioning containers the CFG denoiser unpacks on every step.

Restates modules/prompt_parser.py:136 (ScheduledPromptConditioning), :205-233 (get_multicond_prompt_list: AND splitting and
":weight" suffixes), :239-249 (Composable / Multicond containers), :266-276 (DictWithShape), :278-304 (reconstruct_cond_batch),
:307-318 (stack_conds), :321-349 (reconstruct_multicond_batch).  Pinned by tests/golden/prompt_cond.{npz,json}, which
make_golden produces by executing those parts of the reference file.  The prompt-editing grammar itself (lark) and the text
encoders are not on the path.
"""
import re
from collections import namedtuple

import torch

Scheduled = namedtuple("Scheduled", ["end_at_step", "cond"])
Composable = namedtuple("Composable", ["schedules", "weight"])

_AND = re.compile(r"\bAND\b")
_WEIGHT = re.compile(r"^((?:\s|.)*?)(?:\s*:\s*([-+]?(?:\d+\.?|\d*\.\d+)))?\s*$")


def multicond_prompt_list(prompts):
    """-> (per prompt [(index into the flat list, weight)], flat list of distinct sub-prompts, text -> index)."""
    per_prompt, flat, index_of = [], [], {}
    for prompt in prompts:
        entry = []
        for sub in _AND.split(prompt):
            m = _WEIGHT.search(sub)
            text, weight = m.groups() if m is not None else (sub, 1.0)
            weight = float(weight) if weight is not None else 1.0
            if text not in index_of:
                index_of[text] = len(flat)
                flat.append(text)
            entry.append((index_of[text], weight))
        per_prompt.append(entry)
    return per_prompt, flat, index_of


def _active(schedules, step):
    """The first entry whose end_at_step has not passed; the first one when all have."""
    for k, entry in enumerate(schedules):
        if step <= entry.end_at_step:
            return k
    return 0


def _pad_and_stack(tensors):
    """stack_conds: shorter conds are extended with copies of their last token vector."""
    n = max(t.shape[0] for t in tensors)
    return torch.stack([t if t.shape[0] == n else torch.vstack([t, t[-1:].repeat([n - t.shape[0], 1])]) for t in tensors])


def reconstruct_cond_batch(c, step):
    """c: per image a list of Scheduled -> [B, T, C] tensor (or a dict of such for SDXL's crossattn / vector conds)."""
    first = c[0][0].cond
    picked = [sch[_active(sch, step)].cond for sch in c]
    if isinstance(first, dict):
        return {k: torch.stack([p[k] for p in picked]).to(first[k].dtype) for k in first}
    return torch.stack(picked).to(first.dtype)


def reconstruct_multicond_batch(batch, step):
    """batch: per image a list of Composable -> (conds_list [[(row, weight)]], stacked conds [sum of prompts, T, C])."""
    tensors, conds_list = [], []
    for composable_prompts in batch:
        rows = []
        for cp in composable_prompts:
            rows.append((len(tensors), cp.weight))
            tensors.append(cp.schedules[_active(cp.schedules, step)].cond)
        conds_list.append(rows)
    if isinstance(tensors[0], dict):
        return conds_list, {k: _pad_and_stack([t[k] for t in tensors]) for k in tensors[0]}
    return conds_list, _pad_and_stack(tensors)
