"""Where the Python garbage collector may run while the engine serves.

The reference leaves the collector alone (nanovllm/engine/llm_engine.py has no `gc` call): on its device a step takes tens
of milliseconds.  Here a decode step is 1.4 ms and a request's first token is stamped on the host clock
(bench/serving_bench.py:35-48 of the reference: `first_token_time - submission_time`), so ONE full (generation-2)
collection - 100 ms and more in a process that holds torch, transformers, a model and its captured graphs, all of which
it walks every time - inside `step()` is a TTFT or TPOT outlier of two orders of magnitude.

Policy (`Config.gc_control`, default on):

  * `settle()` after warm-up: one full collection, then `gc.freeze()` - everything alive at that point (modules,
    weights' Python shells, graphs, staging buffers, the imported libraries) is permanent and moves to the permanent
    generation, which no later collection walks;
  * while the engine is inside `step()` the automatic collector is off (`StepGuard`); the young generations are collected
    explicitly at the one point of a step where the host has nothing to do but wait for the device (`slack()`, called
    with the next step already queued), every `young_every` steps generation 0, every `mid_every` generation 1;
  * a full collection runs only when the engine is idle (`idle()`: no request anywhere), over what was allocated since
    `settle()` - the frozen part is never walked again.

The policy changes PROCESS-WIDE collector state from a library (`gc.freeze()`, `gc.disable()` inside a step): an
application that manages the collector itself turns it off with `Config.gc_control = False` (or `MI355_GC_CONTROL=0`,
which the TP workers honour as well).  Several engines in one process share the frozen heap: the freeze is reference-
counted (`_FROZEN_BY`), only the last engine to leave unfreezes, and the automatic collector comes back only when no
engine is inside a step.

`stats` counts what ran where; `watch()` registers a `gc.callbacks` hook that records every collection with its
generation, duration and whether a step was in progress (bench.py reports it, tests assert on it).
"""
from __future__ import annotations

import gc
from time import perf_counter


_FROZEN_BY = 0   # engines (HostGc objects) that have settled and not yet released: the heap stays frozen while > 0
_IN_STEP = 0     # engines currently inside step(): the automatic collector stays off while > 0
MAX_EVENTS = 4096  # watch(): a serving engine runs for days - keep the tail


def freeze_permanent_heap() -> None:
    """One full collection, then freeze what is alive (a TP worker's start-up: ModelRunner.loop)."""
    global _FROZEN_BY
    gc.collect()
    gc.freeze()
    _FROZEN_BY += 1


def release_permanent_heap() -> None:
    global _FROZEN_BY
    _FROZEN_BY = max(0, _FROZEN_BY - 1)
    if _FROZEN_BY == 0:
        gc.unfreeze()


class HostGc:
    def __init__(self, enabled: bool = True, young_every: int = 16, mid_every: int = 512):
        self.enabled = enabled
        self.young_every, self.mid_every = young_every, mid_every
        self.in_step = False
        self._was_enabled = gc.isenabled()
        self._steps = 0
        self._settled = False
        self.stats = {"settle_ms": 0.0, "young": 0, "mid": 0, "full_idle": 0, "frozen_objects": 0}
        self.events: list[dict] = []  # filled by the callback of watch()
        self._t0 = 0.0
        self._watching = False

    # ------------------------------------------------------------------ lifecycle
    def settle(self) -> None:
        """End of start-up: collect once, freeze what is left, take the automatic collector out of the step loop."""
        if not self.enabled:
            return
        t = perf_counter()
        freeze_permanent_heap()
        self.stats["settle_ms"] = (perf_counter() - t) * 1e3
        self.stats["frozen_objects"] = gc.get_freeze_count()
        self._settled = True

    def release(self) -> None:
        """Engine exit: give the process its collector back."""
        if self._watching:
            self.unwatch()
        if not self.enabled or not self._settled:
            return
        self._settled = False
        if self.in_step:
            self.leave_step()
        release_permanent_heap()  # (another engine of this process may still rely on the frozen heap)
        if self._was_enabled and _IN_STEP == 0:
            gc.enable()

    # ------------------------------------------------------------------ the step loop
    def enter_step(self) -> None:
        global _IN_STEP
        if self.enabled and not self.in_step:
            _IN_STEP += 1
            gc.disable()
        self.in_step = True

    def leave_step(self) -> None:
        global _IN_STEP
        if self.enabled and self.in_step:
            _IN_STEP = max(0, _IN_STEP - 1)
            if self._was_enabled and _IN_STEP == 0:
                gc.enable()
        self.in_step = False

    def slack(self) -> None:
        """The host is about to wait for the device with the next step already queued: the cheap collections go here."""
        if not self.enabled:
            return
        self._steps += 1
        if self._steps % self.mid_every == 0:
            gc.collect(1)
            self.stats["mid"] += 1
        elif self._steps % self.young_every == 0:
            gc.collect(0)
            self.stats["young"] += 1

    def idle(self) -> None:
        """No request anywhere in the engine: the only place a full collection is allowed."""
        if self.enabled and self._settled:
            gc.collect()
            self.stats["full_idle"] += 1

    # ------------------------------------------------------------------ observation
    def _callback(self, phase, info):
        if phase == "start":
            self._t0 = perf_counter()
        else:
            if len(self.events) >= MAX_EVENTS:
                del self.events[:MAX_EVENTS // 2]
            self.events.append({"generation": info["generation"], "ms": (perf_counter() - self._t0) * 1e3,
                                "in_step": self.in_step, "collected": info.get("collected", 0),
                                "at": self._t0})

    def watch(self) -> None:
        if not self._watching:
            gc.callbacks.append(self._callback)
            self._watching = True

    def unwatch(self) -> None:
        if self._watching:
            try:
                gc.callbacks.remove(self._callback)
            except ValueError:
                pass
            self._watching = False

    def summary(self, since: float = 0.0, until: float = float("inf")) -> dict:
        ev = [e for e in self.events if since <= e["at"] < until]
        return {"collections": len(ev), "ms": sum(e["ms"] for e in ev),
                "full": sum(e["generation"] == 2 for e in ev),
                "full_in_step": sum(e["generation"] == 2 and e["in_step"] for e in ev),
                "max_ms": max((e["ms"] for e in ev), default=0.0)}
