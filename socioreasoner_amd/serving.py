"""Continuous batching over the engine's batch rows (BASELINE.json configs[2]: "batch=32 continuous batching"; the
reference gets this from vLLM's scheduler behind roll/distributed/strategy/vllm_strategy.py:156-205).

A request is admitted into a free row as soon as one exists (its prompt is prefilled without disturbing running rows),
all rows decode together -- one hipGraph replay per token for the whole batch -- and a row is released the moment its
sequence hits an eos token or its own max_new_tokens.  Greedy decoding; a request's tokens do not depend on what shares
the batch with it (per-row arithmetic is independent of the other rows).
"""
from __future__ import annotations

import contextlib
from collections import deque
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch


# Other GPU work of the same process that the scheduler does not schedule (SAM2's image encoder prefetched on a side stream under stage-1 generation,
# socioreasoner_amd/sam2.py) announces itself here: a cost measurement that overlaps it says nothing about the scheduler's own two streams -- taken anyway, it made
# the share model flip between 4 and 5 CUs per shader engine from run to run of the two-stage pipeline (stage 2: 4.7 s or 5.9 s) -- so none is started while such
# work is running and one in flight when it starts or ends is dropped.
_FOREIGN = {"active": 0, "epoch": 0}


def foreign_gpu_load(begin: bool) -> None:
    """bracket GPU work outside the scheduler's streams: foreign_gpu_load(True) ... foreign_gpu_load(False) (any thread)"""
    _FOREIGN["active"] += 1 if begin else -1
    _FOREIGN["epoch"] += 1


@dataclass
class Request:
    ids: np.ndarray                       # int64 [S], image placeholders already expanded
    pos3: np.ndarray                      # int64 [3, S]
    max_new: int
    images: list = field(default_factory=list)     # uint8 HWC device tensors
    grids: list = field(default_factory=list)      # (t, h, w) per image
    tag: object = None
    aborted: bool = False                 # set by the serving loop: the row runs to its end, its result is discarded


class ContinuousBatcher:
    def __init__(self, engine, eos: Sequence[int], pad_id: int, steps_per_poll: int = 8, sampling: Optional[dict] = None,
                 time_phases: bool = False, overlap: bool = False, admit_cus_per_se="auto"):
        """sampling: None = greedy, else {"temperature", "top_k" (1..1024), "top_p", "seed"} shared by all requests.
        time_phases: bracket every ViT / prefill / decode call with events on the launch stream (phase_ms() sums them).
        overlap: admissions are STAGED into spare KV slots (engine.kv_slots > max_batch) on a CU-masked side stream while the running
        rows keep decoding on the rest of the chip, and committed into rows as they free up (socioreasoner_amd/streams.py).
        admit_cus_per_se: CUs (of 8 per shader engine) of the admission stream, or "auto": chosen per admission so that it ends about
        when the running rows do -- share = ceil(8 A / (A + D)), A = the group's admission time on the whole chip (rate measured on the
        first, unshared admission, scaled by the group's patch / token count), D = the running rows' remaining decode time (step time
        measured on the first chunk); 3 until both are known.  Measured: 448-pixel tiles 2 / 3 / 4 CUs -> 72.3 / 79.3 / 77.1 tiles/s
        (auto: 3); two-image samples 3 / 4 / 5 -> 59.3 / 66.6 / 64.6 samples/s (auto: 4)."""
        self.engine, self.eos, self.pad_id, self.steps_per_poll = engine, [int(e) for e in eos], int(pad_id), steps_per_poll
        self._ev = [] if time_phases else None
        self.overlap = bool(overlap) and engine.kv_slots > engine.cfg.max_batch
        self.staged = None                     # (requests, kv slots, completion event) of the admission in flight
        self.row_slot: Dict[int, int] = {}
        self.free_slots = deque(range(engine.kv_slots))
        self._dec_last = None
        self._commit_ev = None
        self._deferred: list = []               # (requests, token counts, device token rows, event) of rows that finished, not yet reported (_deliver)
        self._copy_stream = None
        import os
        if admit_cus_per_se == "auto" and os.environ.get("SR_ADMIT_CUS"):       # deployment / experiment override of the automatic share (2 .. 6, fractions allowed)
            admit_cus_per_se = float(os.environ["SR_ADMIT_CUS"])
        self._auto = admit_cus_per_se == "auto"
        self._share = 3 if self._auto else admit_cus_per_se
        # calibration of the "auto" share: ms per work unit of an unshared admission, ms per decode step with the chip to itself.  Kept on
        # the engine, so that the next scheduler on it (one per generate call) starts calibrated
        self._adm_rate, self._step_ms = getattr(engine, "_sched_cal", (None, None))
        self._cal_adm = self._cal_step = None      # (start event, end event, units / steps) of a measurement in flight
        # round 5: the cost of SHARING the chip depends on the rows per step (a 128-row decode step is 2.1 x as long next to an
        # admission, a 32-row one 1.27 x), so the two tables below are only the prior: every shared decode chunk and every shared admission is timed and the
        # measured slow-down of the share in use replaces the table's entry (other shares keep the table's shape, scaled by what was measured).  Kept on the
        # engine like the unshared calibration.
        self._dec_meas, self._adm_meas = getattr(engine, "_sched_cal_shared", ({}, {}))
        self._cal_dec_sh = self._cal_adm_sh = None
        self._meas_epoch: Dict[str, int] = {}      # measurement in flight -> _FOREIGN epoch at its start (see foreign_gpu_load)
        self._cal_seen = engine.__dict__.setdefault("_sched_cal_seen", {})      # ("dec" | "adm", share) -> samples taken (8 fast ones per share, then every 8th opportunity)
        self._cal_opp: Dict[tuple, int] = {}
        import os
        self._online = os.environ.get("SR_SCHED_ONLINE", "1") != "0"      # 0: the 32-row tables only (A/B hook of tools/gpu_lease.sh sched_ab)
        self._cnt_last: Dict[int, int] = {}
        if self.overlap:
            from .streams import overlap_streams
            self.streams = overlap_streams(engine.device, self._share)
            c = engine.cfg     # relative cost of a patch (ViT) and of a prompt token (LM prefill): their linear-layer parameter counts
            self._wp = c.v_depth * (4 * c.v_hidden * c.v_hidden + 3 * c.v_hidden * c.v_inter)
            self._wt = c.t_layers * (c.t_hidden * (2 * c.t_heads + 2 * c.t_kv_heads) * 128 + 3 * c.t_hidden * c.t_inter)
        self.free = deque(range(engine.cfg.max_batch))
        self.active: Dict[int, Request] = {}
        self.pending: deque = deque()
        # rounds / host_ms / poll_wait_ms (round 5): scheduling rounds, host time spent in them OUTSIDE the poll's wait for the device, and that wait --
        # with N ranks on one node a flat 1 -> N curve shows here as host_ms per round growing with N (host contention) rather than poll_wait_ms
        self.stats = {"admitted": 0, "steps": 0, "admissions": 0, "staged_shared": 0, "steps_shared": 0, "shares": [], "rounds": 0, "host_ms": 0.0, "poll_wait_ms": 0.0}
        engine.rows_begin()
        if sampling:
            engine.rows_sampling(float(sampling["temperature"]), int(sampling["top_k"]), float(sampling.get("top_p", 1.0)), int(sampling.get("seed", 0)))

    def _mark(self):
        if self._ev is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.engine.device))
        return e

    def _span(self, name, a, b):
        if self._ev is not None:
            self._ev.append((name, a, b))

    def phase_ms(self) -> Dict[str, float]:
        """Sum of the event-bracketed spans per phase (synchronises)."""
        out = {"vit": 0.0, "prefill": 0.0, "decode": 0.0}
        if self.overlap:      # spans that shared the chip (admission on its CU set under decode on the rest) are kept apart
            out.update({"vit_shared": 0.0, "prefill_shared": 0.0, "decode_shared": 0.0})
        if self._ev:
            torch.cuda.synchronize(self.engine.device)
            for name, a, b in self._ev:
                out[name] += a.elapsed_time(b)
        return out

    def submit(self, req: Request):
        """A request that can never be admitted (more prompt tokens / patches than ONE admission may carry, or no room for a single new
        token) is refused here: inside the pump it would take KV slots with it when the engine rejects the group."""
        cfg = self.engine.cfg
        npatch = sum(t * h * w for t, h, w in req.grids)
        if len(req.ids) > cfg.max_prefill_tokens or npatch > cfg.max_patches or len(req.ids) + 1 > cfg.max_ctx:
            raise ValueError(f"request of {len(req.ids)} tokens / {npatch} patches exceeds the engine's capacity "
                             f"(max_prefill_tokens {cfg.max_prefill_tokens}, max_patches {cfg.max_patches}, max_ctx {cfg.max_ctx})")
        self.pending.append(req)

    def idle(self) -> bool:
        return not self.pending and not self.active and self.staged is None and not self._deferred

    def abort(self, match: Callable[[Request], bool]) -> int:
        """ABORT (reference vllm_strategy.py:188-193 -> vLLM abort_request): queued requests are dropped, running rows stop NOW and
        give row and KV slot back at the next poll (sr_rows_abort), requests that are prefilled but still wait for a row are stopped the
        moment they are installed.  Their results are discarded (Request.aborted).  Returns the number of requests hit."""
        n0 = len(self.pending)
        self.pending = deque(r for r in self.pending if not match(r))
        hit = n0 - len(self.pending)
        rows = [row for row, r in self.active.items() if match(r) and not r.aborted]
        for row in rows:
            self.active[row].aborted = True
        if rows:
            s = self._dec_last if (self.overlap and self._dec_last is not None) else torch.cuda.current_stream(self.engine.device)
            with torch.cuda.stream(s):              # ordered between two decode chunks of the rows' own stream
                self.engine.rows_abort(rows)
        for r in (self.staged[0] if self.staged is not None else []):
            if match(r) and not r.aborted:
                r.aborted = True
                hit += 1
        return hit + len(rows)

    def _admit(self):
        cfg = self.engine.cfg
        grp, ntok, npatch = [], 0, 0
        while self.pending and len(grp) < len(self.free):
            r = self.pending[0]
            np_r = sum(t * h * w for t, h, w in r.grids)
            if grp and (ntok + len(r.ids) > cfg.max_prefill_tokens or npatch + np_r > cfg.max_patches):
                break
            grp.append(self.pending.popleft())
            ntok += len(r.ids)
            npatch += np_r
        if not grp:
            return
        rows = [self.free.popleft() for _ in grp]
        emb = None
        ims = [im for r in grp for im in r.images]
        t0 = self._mark()
        try:
            if ims:
                pix = torch.cat([self.engine.patchify(im) for im in ims], dim=0)
                emb = self.engine.vit_forward(pix, [g for r in grp for g in r.grids])
            t1 = self._mark()
            self.engine.admit(rows, [r.ids for r in grp], [r.pos3 for r in grp], [r.max_new for r in grp], emb)
        except BaseException:
            self.free.extendleft(reversed(rows))
            raise
        self._span("vit", t0, t1)
        self._span("prefill", t1, self._mark())
        for row, r in zip(rows, grp):
            self.active[row] = r
        self.stats["admitted"] += len(grp)
        self.stats["admissions"] += 1

    # ------------------------------------------------------------------ overlapped admission (stage on a side stream, commit into rows)
    def _take_group(self, limit: int):
        cfg = self.engine.cfg
        grp, ntok, npatch = [], 0, 0
        while self.pending and len(grp) < limit:
            r = self.pending[0]
            np_r = sum(t * h * w for t, h, w in r.grids)
            if grp and (ntok + len(r.ids) > cfg.max_prefill_tokens or npatch + np_r > cfg.max_patches):
                break
            grp.append(self.pending.popleft())
            ntok += len(r.ids)
            npatch += np_r
        return grp

    def _set_cus(self, n):
        """tell the engine how many CUs the decode steps queued next (on the current stream) will find: its x-stationary down-projection deals
        its weight tiles to that many blocks (sr_rows_set_cus; a hint -- the tokens do not depend on it)"""
        if getattr(self.engine, "_decode_cus", 0) != n:
            self.engine.rows_set_cus(n)
            self.engine._decode_cus = n

    def _use_decode_stream(self, s):
        """decode moves between the unmasked stream and the masked one; the new stream waits for what the old one has queued"""
        if self._dec_last is None:                   # first use: whatever the caller queued (weights, inputs) comes first
            s.wait_stream(torch.cuda.current_stream(self.engine.device))
        elif self._dec_last is not s:
            s.wait_stream(self._dec_last)
        self._dec_last = s
        return s

    # measured on MI355X (bench.py --admit-cus 2 / 3 / 4): an admission confined to c of the 8 CUs of every shader engine takes
    # (8 / c) x _ADM_EFF[c] as long as on the whole chip, a decode step on the other 8 - c CUs _DEC_SLOW[c] as long
    # (6 of 8: round 6, `bench.py --pair --tile 896 --admit-cus 6` -- admissions of 822 ms on the whole chip against 421 ms of decode per wave: 29.9 samples/s
    # against 27.4 on 5 CUs, profiles/r06_pair896_share_sweep.txt; the decode step on the remaining 64 CUs is 2.2 x as long)
    _ADM_EFF = {2: 0.86, 3: 0.886, 4: 0.913, 5: 0.94, 6: 0.955}
    _DEC_SLOW = {2: 1.20, 3: 1.265, 4: 1.36, 5: 1.55, 6: 2.2}

    def _dec_factor(self, c: int) -> float:
        """decode step time on the 8 - c CUs per shader engine next to an admission, relative to the whole chip"""
        if c in self._dec_meas:
            return self._dec_meas[c]
        if self._dec_meas:       # the table's shape, scaled by the measured excess of the shares that were seen
            k = sum((v - 1.0) / (self._DEC_SLOW[m] - 1.0) for m, v in self._dec_meas.items()) / len(self._dec_meas)
            return 1.0 + (self._DEC_SLOW[c] - 1.0) * max(k, 0.25)
        return self._DEC_SLOW[c]

    def _adm_factor(self, c: int) -> float:
        """admission time on c of the 8 CUs per shader engine under decode, relative to the whole chip"""
        if c in self._adm_meas:
            return self._adm_meas[c]
        base = (8.0 / c) * self._ADM_EFF[c]
        if self._adm_meas:
            k = sum(v / ((8.0 / m) * self._ADM_EFF[m]) for m, v in self._adm_meas.items()) / len(self._adm_meas)
            return base * min(max(k, 0.5), 2.0)
        return base

    def _meas_begin(self, key: str) -> bool:
        """may a cost measurement start now?  Not next to GPU work the scheduler does not own."""
        if _FOREIGN["active"] > 0:
            return False
        self._meas_epoch[key] = _FOREIGN["epoch"]
        return True

    def _meas_clean(self, key: str) -> bool:
        """did the measurement run without such work starting or ending under it?"""
        return self._meas_epoch.pop(key, None) == _FOREIGN["epoch"]

    def _ema(self, table: dict, key: int, value: float):
        """fast while a share is new (the first 8 samples), slow afterwards: the measurement never stops (a table frozen after 8 samples keeps
        a stale factor when the workload on a long-lived engine changes -- prompt lengths, context, rows per step)"""
        kind = "dec" if table is self._dec_meas else "adm"
        a = 0.5 if self._cal_seen.get((kind, key), 0) < 8 else 0.125
        table[key] = value if key not in table else (1.0 - a) * table[key] + a * value

    def _sample_due(self, kind: str) -> bool:
        """every opportunity for a share's first 8 samples, every 8th afterwards (two timed events per sample)"""
        n = self._cal_seen.get((kind, self._share), 0)
        if n < 8:
            return True
        k = self._cal_opp[(kind, self._share)] = self._cal_opp.get((kind, self._share), 0) + 1
        return k % 8 == 0

    def _pick_share(self, a_ms: float, steps_left: float) -> int:
        """CUs per shader engine for an admission that takes a_ms on the whole chip while the running rows still have steps_left decode
        steps in front of them: the share with the shortest predicted time until those rows are done AND the admission has landed
        (decode runs on the small CU set until the poll after the admission ends, on the whole chip afterwards).  Among the shares predicted within 1.5 % of the
        best the smallest is taken (it leaves decode more of the chip); the band was 4 % while the costs came from the 32-row table alone -- with
        the costs measured on the engine itself the prediction is trusted further (64 rows: 3 CUs predicted 2 % behind 4, measured 1.4 % behind)."""
        t = {}
        for c in (2, 3, 4, 5, 6):
            ta = a_ms * self._adm_factor(c)
            sc = self._step_ms * self._dec_factor(c)
            shared = -(-(ta / sc) // self.steps_per_poll) * self.steps_per_poll      # steps decoded next to the admission (whole chunks)
            # an admission that outlasts the running rows leaves the rest of the chip idle beside its CU mask until it lands: such a plan is
            # charged 5 % more than its own length (64 rows: share 3 predicted 467 ms against 471 ms for share 4, measured 1.4 % SLOWER --
            # the admission was the critical path, and every per cent it overran its estimate was a per cent of idle decode)
            t[c] = 1.05 * ta if shared >= steps_left else shared * sc + (steps_left - shared) * self._step_ms
        t_min = min(t.values())
        return min(c for c in t if t[c] <= 1.015 * t_min)

    # Measured and dropped (round 3): cutting a staged group to the rows that are idle or about to finish (so that they wait for a short
    # admission instead of a full one; the next group cannot be staged before this one is fully installed) -- ragged phase of bench.py,
    # 128 requests through 32 rows: 54.6 tiles/s / 768 decode steps with groups of >= 4, 56.7 / 744 with >= 8, against 59.7 / 720 with
    # groups that fill every spare KV slot: many small admissions cost more (their GEMMs, and the decode steps slowed beside them) than
    # the waiting they save.
    def _stage(self):
        grp = self._take_group(min(len(self.free_slots), self.engine.cfg.max_batch))
        if not grp:
            return
        slots = [self.free_slots.popleft() for _ in grp]
        units = sum(self._wp * sum(t * h * w for t, h, w in r.grids) + self._wt * len(r.ids) for r in grp)
        # with rows running: the admission's half of the chip; with nothing to decode there is nothing to share the chip with
        shared = "_shared" if self.active else ""
        if self.active:
            if self._auto and self._adm_rate is not None and self._step_ms is not None:
                left = [max(r.max_new - self._cnt_last.get(row, 0), 1) for row, r in self.active.items()]
                share = self._pick_share(units * self._adm_rate, sum(left) / len(left))
                if share != self._share:
                    from .streams import overlap_streams
                    self._share, self.streams = share, overlap_streams(self.engine.device, share)
            self.stats["shares"].append(self._share)
            s = self.streams.admit
            s.wait_stream(torch.cuda.current_stream(self.engine.device))     # the request's images were produced on the caller's stream
            self.stats["staged_shared"] += len(grp)
            if self._commit_ev is not None:
                s.wait_event(self._commit_ev)          # the previous group's last commit reads the engine's admission scratch
        else:
            s = self._use_decode_stream(self.streams.decode_full)
        cal = bool(self._auto and not shared and self._adm_rate is None and self._cal_adm is None and self._meas_begin("adm"))
        cal_sh = bool(self._online and self._auto and shared and self._adm_rate is not None and self._cal_adm_sh is None
                      and self._sample_due("adm") and self._meas_begin("adm_sh"))
        try:
            self._stage_on(s, grp, slots, shared, cal or cal_sh, units)
            if cal_sh and self._cal_adm is not None:          # (_stage_on left the event pair in _cal_adm: this one measures a SHARED admission)
                self._cal_adm_sh, self._cal_adm = self._cal_adm + (self._share,), None
        except BaseException:
            self.free_slots.extendleft(reversed(slots))          # the group is lost to its caller (the exception says so), the slots are not
            raise

    def _stage_on(self, s, grp, slots, shared, cal, units):
        with torch.cuda.stream(s):
            if cal:
                c0 = torch.cuda.Event(enable_timing=True)
                c0.record(s)
            t0 = self._mark()
            emb = None
            ims = [im for r in grp for im in r.images]
            if ims:
                pix = torch.cat([self.engine.patchify(im) for im in ims], dim=0)
                emb = self.engine.vit_forward(pix, [g for r in grp for g in r.grids])
            t1 = self._mark()
            self.engine.admit_stage(slots, [r.ids for r in grp], [r.pos3 for r in grp], [r.max_new for r in grp], emb)
            self._span("vit" + shared, t0, t1)
            self._span("prefill" + shared, t1, self._mark())
            ev = torch.cuda.Event(enable_timing=cal)
            ev.record(s)
            if cal:
                self._cal_adm = (c0, ev, units)
        self.staged = (grp, slots, ev, emb)            # emb kept alive until the side stream is done with it
        self.stats["admissions"] += 1

    def _commit(self):
        """install as many staged sequences as there are free rows (in staging order); the rest waits for the next rows"""
        grp, slots, ev, _ = self.staged
        k = min(len(self.free), len(grp))
        rows = [self.free.popleft() for _ in range(k)]
        s = self._use_decode_stream(self.streams.decode_full)
        s.wait_event(ev)
        with torch.cuda.stream(s):
            self.engine.admit_commit(rows)
            self._commit_ev = torch.cuda.Event()
            self._commit_ev.record(s)
        for row, slot, r in zip(rows, slots[:k], grp[:k]):
            self.active[row] = r
            self.row_slot[row] = slot
            self._cnt_last[row] = 0
        dead = [row for row, r in zip(rows, grp[:k]) if r.aborted]      # aborted while they waited for a row: stop before the first step
        if dead:
            with torch.cuda.stream(s):
                self.engine.rows_abort(dead)
        self.stats["admitted"] += k
        del grp[:k], slots[:k]
        if not grp:
            self.staged = None

    def _pump_overlap(self, on_complete):
        if self.staged is not None and self.free and (self.staged[2].query() or not self.active):
            self._commit()                             # (ordered after the staging stream through the event, not on the host)
        if not self.active and self.staged is None and self.pending:
            self._stage()                              # nothing to decode meanwhile: whole chip, install at once
            self._commit()
        if not self.active:
            if self.staged is not None:
                self.staged[2].synchronize()
            return
        busy = self.staged is not None and not self.staged[2].query()
        # The chunk under which the next admission gets staged is queued on the UNMASKED stream: its steps_per_poll steps own every CU and the
        # admission's first GEMMs queue behind them.  Putting that chunk on the decode CU set instead (the admission starts at once, the
        # chunk decodes on 160 CUs) was measured and is slower -- 76.8 vs 78.3 tiles/s, two runs each, same box: the admission is not on
        # the critical path for that long, the decode rows are.
        s = self._use_decode_stream(self.streams.decode if busy else self.streams.decode_full)
        # bookkeeping only: a chunk under which the next admission gets staged shares the chip with it from the moment the admission's first
        # kernel starts (round 3 counted it as "alone", which made the unshared step look 6 % slower than the static one: tools/probe_rows_step.py)
        shares = busy or (self.staged is None and bool(self.pending) and bool(self.free_slots))
        # (step-time calibration: a chunk with the chip to itself -- no admission in flight and none about to be staged under it)
        cal = (self._auto and not busy and not (self.staged is None and self.pending and self.free_slots)
               and self._step_ms is None and self._cal_step is None and self._meas_begin("step"))
        cal_sh = bool(self._online and self._auto and busy and self._step_ms is not None and self._cal_dec_sh is None       # a chunk on the decode CU set next to the admission
                      and self._sample_due("dec") and self._meas_begin("dec_sh"))
        with torch.cuda.stream(s):
            if cal or cal_sh:
                c0 = torch.cuda.Event(enable_timing=True)
                c0.record(s)
            t0 = self._mark()
            self._set_cus(self.streams.decode_cus if s is self.streams.decode else 0)
            self.engine.rows_step(self.steps_per_poll, self.eos, self.pad_id)
            self._span("decode_shared" if shares else "decode", t0, self._mark())
            if cal or cal_sh:
                c1 = torch.cuda.Event(enable_timing=True)
                c1.record(s)
                if cal:
                    self._cal_step = (c0, c1, self.steps_per_poll)
                else:
                    self._cal_dec_sh = (c0, c1, self.steps_per_poll, self._share, self.staged[2])
        self.stats["steps"] += self.steps_per_poll
        self.stats["steps_shared"] += self.steps_per_poll if shares else 0
        self._deliver(on_complete)                     # the rows that finished in the previous round (their callbacks may submit new requests)
        if self.staged is None and self.pending and self.free_slots:
            self._stage()                              # the host side of the next admission is prepared while the chunk above runs
        with torch.cuda.stream(s):
            fin, cnt = self._poll()
            self._cnt_last = {row: int(cnt[row]) for row in self.active}
            if self._auto:          # (the poll synchronised the decode stream: finished measurements can be read without waiting)
                if self._cal_step is not None and self._cal_step[1].query():
                    if self._meas_clean("step"):
                        self._step_ms = self._cal_step[0].elapsed_time(self._cal_step[1]) / self._cal_step[2]
                        self.engine._sched_cal = (self._adm_rate, self._step_ms)
                    self._cal_step = None
                if self._cal_adm is not None and self._cal_adm[1].query():
                    if self._meas_clean("adm"):
                        self._adm_rate = self._cal_adm[0].elapsed_time(self._cal_adm[1]) / max(self._cal_adm[2], 1)
                        self.engine._sched_cal = (self._adm_rate, self._step_ms)
                    self._cal_adm = None
                if self._cal_dec_sh is not None and self._cal_dec_sh[1].query():
                    c0, c1, n_st, share, adm_ev = self._cal_dec_sh
                    if not adm_ev.query() and self._meas_clean("dec_sh"):          # the admission outlasted the chunk: every one of its steps shared the chip
                        self._ema(self._dec_meas, share, max(c0.elapsed_time(c1) / n_st / self._step_ms, 1.0))
                        self._cal_seen[("dec", share)] = self._cal_seen.get(("dec", share), 0) + 1
                        self.engine._sched_cal_shared = (self._dec_meas, self._adm_meas)
                    self._cal_dec_sh = None
                if self._cal_adm_sh is not None and self._cal_adm_sh[1].query():
                    c0, ev, units, share = self._cal_adm_sh
                    if self._meas_clean("adm_sh"):
                        self._ema(self._adm_meas, share, max(c0.elapsed_time(ev) / max(units * self._adm_rate, 1e-6), 1.0))
                        self._cal_seen[("adm", share)] = self._cal_seen.get(("adm", share), 0) + 1
                        self.engine._sched_cal_shared = (self._dec_meas, self._adm_meas)
                    self._cal_adm_sh = None
                self.stats["share_model"] = {"decode_slowdown_measured": {k: round(v, 3) for k, v in self._dec_meas.items()},
                                             "admission_slowdown_measured": {k: round(v, 3) for k, v in self._adm_meas.items()},
                                             "step_ms": self._step_ms, "admission_ms_per_unit": self._adm_rate}
            # finished rows: their tokens are copied out in stream order (device to device) and the rows are free at once; the host copy and the
            # callbacks wait until the NEXT chunk of decode steps has been queued (_deliver) -- a wave of 32 rows ending together used to cost 32
            # synchronous read-backs (~5 ms) between two chunks, with the rows' stream idle
            rows = [r for r in self.active if fin[r]]
            if rows:
                counts = [int(cnt[r]) for r in rows]
                buf = self.engine.rows_tokens(rows, counts)
                ev = torch.cuda.Event()
                ev.record(s)
                self._deferred.append(([self.active.pop(r) for r in rows], counts, buf, ev))
                for row in rows:
                    self.free.append(row)
                    self.free_slots.append(self.row_slot.pop(row))
            if not self.active:          # nothing will be queued behind them: deliver now
                self._deliver(on_complete)

    def _deliver(self, on_complete):
        """host copies + callbacks of the rows that finished in earlier rounds (on a copy stream that waits for the read-out only, not for the decode
        steps queued since)"""
        if not self._deferred:
            return
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.engine.device)
        todo, self._deferred = self._deferred, []
        for reqs, counts, buf, ev in todo:
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(ev)
                buf.record_stream(self._copy_stream)
                host = buf.cpu().numpy()
            # the callbacks run with the rows' stream current, as they did when they were called from the poll: device work they queue (the bench's
            # raster tail and result exchange) is ordered behind the decode steps and never lands on the null stream, which would synchronise with the
            # CU-masked (blocking) streams -- i.e. wait for a staged admission and hold every later decode chunk behind itself
            with (torch.cuda.stream(self._dec_last) if (self.overlap and self._dec_last is not None) else contextlib.nullcontext()):
                for req, n, toks in zip(reqs, counts, host):
                    on_complete(req, toks[:n].tolist())

    def _poll(self):
        import time
        t0 = time.perf_counter()
        out = self.engine.rows_poll()
        self._poll_wait += time.perf_counter() - t0
        return out

    def pump(self, on_complete: Callable[[Request, List[int]], None]):
        """One scheduling round: admit what fits, decode `steps_per_poll` tokens, release finished rows."""
        import time
        t0 = time.perf_counter()
        self._poll_wait = 0.0
        try:
            return self._pump(on_complete)
        finally:
            dt = time.perf_counter() - t0
            self.stats["rounds"] += 1
            self.stats["poll_wait_ms"] += self._poll_wait * 1e3
            self.stats["host_ms"] += (dt - self._poll_wait) * 1e3

    def _pump(self, on_complete):
        if self.overlap:
            return self._pump_overlap(on_complete)
        self._admit()
        if not self.active:
            return
        t0 = self._mark()
        self.engine.rows_step(self.steps_per_poll, self.eos, self.pad_id)
        self._span("decode", t0, self._mark())
        self.stats["steps"] += self.steps_per_poll
        fin, cnt = self._poll()
        for row in [r for r in self.active if fin[r]]:
            req = self.active.pop(row)
            toks = self.engine.row_tokens(row, int(cnt[row])).cpu().tolist()
            self.free.append(row)
            on_complete(req, toks)

    def _hand_back(self):
        """back to the caller's stream, the engine's CU hint back to the whole chip"""
        if self.overlap and self._dec_last is not None:
            torch.cuda.current_stream(self.engine.device).wait_stream(self._dec_last)
            self._set_cus(0)

    def run_stream(self, requests: Sequence[Request], on_complete: Callable[[Request, List[int]], None]) -> None:
        """Serve a stream of requests, reporting each as it completes (in completion order): what a server does -- the next requests'
        admission is staged under the running rows' last decode steps instead of after them."""
        for r in requests:
            self.submit(r)
        while not self.idle():
            self.pump(on_complete)
        self._hand_back()

    def run(self, requests: Sequence[Request]) -> List[List[int]]:
        """Serve a fixed list of requests; returns their token lists in request order."""
        order = {id(r): i for i, r in enumerate(requests)}
        out: List[Optional[List[int]]] = [None] * len(requests)
        for r in requests:
            self.submit(r)
        while not self.idle():
            self.pump(lambda req, toks: out.__setitem__(order[id(req)], toks))
        self._hand_back()
        return out
