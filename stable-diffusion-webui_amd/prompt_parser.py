"""Conditioning containers between the text encoder and the CFG denoiser — host mirror of the parts of
modules/prompt_parser.py the sampling loop touches every step:

  ScheduledPromptConditioning (:136), SdConditioning (:139-154), get_learned_conditioning (:158-202),
  get_multicond_prompt_list (:205-233), ComposableScheduledPromptConditioning / MulticondLearnedConditioning (:239-249),
  get_multicond_learned_conditioning (:252-266), DictWithShape (:266-276), reconstruct_cond_batch (:278-304), stack_conds
  (:307-318), reconstruct_multicond_batch (:321-349).

Same names, arguments and results.  The prompt-editing grammar ("[from:to:when]", alternation; lark, :9-133) is the webui's
parser: ``get_learned_conditioning`` takes its output through ``prompt_schedules`` and treats prompts literally without it.
Only data movement happens here (index selection, stacking, padding by repeating the last token vector).
"""
from __future__ import annotations

import re
from collections import namedtuple

import torch

ScheduledPromptConditioning = namedtuple("ScheduledPromptConditioning", ["end_at_step", "cond"])


class SdConditioning(list):
    """A list of prompts that also carries is_negative_prompt and the image size SDXL's conditioner needs (:139-154)."""

    def __init__(self, prompts, is_negative_prompt=False, width=None, height=None, copy_from=None):
        super().__init__()
        self.extend(prompts)
        if copy_from is None:
            copy_from = prompts
        self.is_negative_prompt = is_negative_prompt or getattr(copy_from, 'is_negative_prompt', False)
        self.width = width or getattr(copy_from, 'width', None)
        self.height = height or getattr(copy_from, 'height', None)


def get_learned_conditioning(model, prompts, steps, hires_steps=None, use_old_scheduling=False, prompt_schedules=None):
    """:158-202.  ``model.get_learned_conditioning(texts)`` encodes a list of texts to [n, T, C] (or a dict of tensors).
    ``prompt_schedules``: per prompt [[end_at_step, text], ...] as get_learned_conditioning_prompt_schedules returns (:25-133);
    None = no prompt editing, one entry lasting all ``steps``.  Equal prompts share one encoded schedule."""
    if prompt_schedules is None:
        prompt_schedules = [[[steps, prompt]] for prompt in prompts]
    encoded = {}
    for prompt, schedule in zip(prompts, prompt_schedules):
        if prompt in encoded:
            continue
        conds = model.get_learned_conditioning(SdConditioning([text for _, text in schedule], copy_from=prompts))
        row = (lambda i: {k: v[i] for k, v in conds.items()}) if isinstance(conds, dict) else (lambda i: conds[i])
        encoded[prompt] = [ScheduledPromptConditioning(end, row(i)) for i, (end, _) in enumerate(schedule)]
    return [encoded[prompt] for prompt in prompts]


re_AND = re.compile(r"\bAND\b")
re_weight = re.compile(r"^((?:\s|.)*?)(?:\s*:\s*([-+]?(?:\d+\.?|\d*\.\d+)))?\s*$")


def get_multicond_prompt_list(prompts):
    """:205-233: split every prompt at AND, read the optional ":weight" suffix, deduplicate the sub-prompts.
    -> (per prompt [(index into the flat list, weight)], flat SdConditioning of distinct sub-prompts, text -> index)."""
    flat = SdConditioning(prompts)
    flat.clear()
    index_of, per_prompt = {}, []
    for prompt in prompts:
        entry = []
        for sub in re_AND.split(prompt):
            m = re_weight.search(sub)
            text, weight = m.groups() if m is not None else (sub, 1.0)
            if text not in index_of:
                index_of[text] = len(flat)
                flat.append(text)
            entry.append((index_of[text], 1.0 if weight is None else float(weight)))
        per_prompt.append(entry)
    return per_prompt, flat, index_of


class ComposableScheduledPromptConditioning:
    def __init__(self, schedules, weight=1.0):
        self.schedules = schedules
        self.weight = weight


class MulticondLearnedConditioning:
    def __init__(self, shape, batch):
        self.shape = shape                                    # the shape field is needed to send this object to DDIM/PLMS
        self.batch = batch


def get_multicond_learned_conditioning(model, prompts, steps, hires_steps=None, use_old_scheduling=False, prompt_schedules=None):
    """:252-266"""
    res_indexes, prompt_flat_list, prompt_indexes = get_multicond_prompt_list(prompts)
    learned_conditioning = get_learned_conditioning(model, prompt_flat_list, steps, hires_steps, use_old_scheduling, prompt_schedules)
    res = [[ComposableScheduledPromptConditioning(learned_conditioning[i], weight) for i, weight in indexes] for indexes in res_indexes]
    return MulticondLearnedConditioning(shape=(len(prompts),), batch=res)


class DictWithShape(dict):
    def __init__(self, x, shape=None):
        super().__init__()
        self.update(x)

    @property
    def shape(self):
        return self["crossattn"].shape


def _target_index(schedules, current_step):
    for current, entry in enumerate(schedules):
        if current_step <= entry.end_at_step:
            return current
    return 0


def reconstruct_cond_batch(c, current_step):
    """:278-304: for every image the schedule entry active at ``current_step`` -> [B, T, C] (dict conds: one tensor per key)."""
    like = c[0][0].cond
    chosen = [schedule[_target_index(schedule, current_step)].cond for schedule in c]
    if isinstance(like, dict):
        batch = {k: torch.stack([x[k] for x in chosen]).to(device=ref.device, dtype=ref.dtype) for k, ref in like.items()}
        return DictWithShape(batch, (len(c),) + like['crossattn'].shape)
    return torch.stack(chosen).to(device=like.device, dtype=like.dtype)


def stack_conds(tensors):
    """:307-318: prompts of different token counts — the shorter ones are extended with copies of their last token vector."""
    longest = max(t.shape[0] for t in tensors)
    return torch.stack([t if t.shape[0] == longest else torch.vstack([t, t[-1:].repeat([longest - t.shape[0], 1])]) for t in tensors])


def reconstruct_multicond_batch(c: MulticondLearnedConditioning, current_step):
    """:321-349 -> (conds_list: per image [(row of the stacked tensor, weight)], stacked conds of every AND-ed sub-prompt)."""
    like = c.batch[0][0].schedules[0].cond
    rows, conds_list = [], []
    for image_prompts in c.batch:
        entry = []
        for sub in image_prompts:
            entry.append((len(rows), sub.weight))
            rows.append(sub.schedules[_target_index(sub.schedules, current_step)].cond)
        conds_list.append(entry)
    if isinstance(rows[0], dict):
        stacked = {k: stack_conds([x[k] for x in rows]) for k in rows[0].keys()}
        return conds_list, DictWithShape(stacked, stacked['crossattn'].shape)
    return conds_list, stack_conds(rows).to(device=like.device, dtype=like.dtype)


def is_multicond(c):
    """True for a MulticondLearnedConditioning — this module's or the webui's own class (modules/prompt_parser.py:245-249):
    inside the webui the conds arrive as the reference's objects, which have the same attributes."""
    return isinstance(c, MulticondLearnedConditioning) or (hasattr(c, "batch") and hasattr(c, "shape") and not torch.is_tensor(c))


def selection_key(c, current_step):
    """Identity of what reconstruct_* would pick at this step: equal keys = identical tensors, so the denoiser can keep the
    reconstructed batch (and the cross-attention K / V projected from it) instead of rebuilding them every step."""
    if is_multicond(c):
        return tuple(tuple((_target_index(cp.schedules, current_step), id(cp.schedules), cp.weight) for cp in img) for img in c.batch)
    return tuple((_target_index(sch, current_step), id(sch)) for sch in c)


def slice_conds(c, lo, hi, device):
    """The conds of images [lo, hi) of a job: a ready tensor [N, T, C] (moved to ``device``), or the containers above —
    MulticondLearnedConditioning (p.c) / a list of per-image schedules (p.uc) — which the CFG denoiser unpacks on every step."""
    if is_multicond(c):
        return MulticondLearnedConditioning((hi - lo,), c.batch[lo:hi])
    if isinstance(c, list):
        return c[lo:hi]
    return c[lo:hi].to(device)
