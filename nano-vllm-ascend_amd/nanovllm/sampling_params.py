"""Per-request sampling knobs (reference: nanovllm/sampling_params.py:5-11)."""
from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False
    # Extension over the reference (which only has multinomial sampling and rejects
    # temperature -> 0): deterministic argmax decoding, needed for parity runs
    # (BASELINE config 1 "greedy decode").  Default keeps the reference behaviour.
    greedy: bool = False

    def __post_init__(self):
        assert self.temperature > 1e-10, "greedy sampling is not permitted (pass greedy=True for argmax)"
