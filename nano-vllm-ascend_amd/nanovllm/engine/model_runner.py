"""Device-side executor of one TP rank (reference: nanovllm/engine/model_runner.py).

Role for role the same as the reference class — build + load the model, size and
allocate the paged KV cache, turn scheduled sequences into device metadata, run the
model eagerly (prefill) or by replaying a captured device graph (decode), sample on
rank 0 — re-designed for one MI355X per process:

  * step metadata is assembled on the host by engine/batch_meta.py (bit-exact with the
    reference's prepare_* lists) directly into ONE pinned staging buffer and reaches the
    GPU with ONE async copy per decode step (the reference does 5-6, :354-357,:234);
  * decode steps are hipGraphs captured per batch bucket (1,2,4,...,max_num_seqs) with
    torch.cuda.CUDAGraph on the stream the C-ABI kernels are enqueued on — the
    counterpart of the torchair "reduce-overhead" capture (:150-154).  Padded rows use
    context_len 0 and the dummy slot in the reserved last block exactly as :303-311;
  * tensor-parallel ranks are separate processes joined by an RCCL ("nccl") group;
    rank 0 publishes each step to the workers through a shared-memory seqlock carrying
    compact int64 arrays (engine/rpc.py) instead of a pickled Sequence list (:172-187).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from nanovllm import ops
from nanovllm.config import Config
from nanovllm.engine import batch_meta
from nanovllm.engine.rpc import StepChannel
from nanovllm.engine.sequence import Sequence
from nanovllm.layers.sampler import Sampler
from nanovllm.models.models_map import model_dict
from nanovllm.utils.context import reset_context, set_context
from nanovllm.utils.loader import has_checkpoint, init_synthetic_weights, load_model


def _torch_dtype_of(hf_config) -> torch.dtype:
    text = getattr(hf_config, "text_config", hf_config)
    dt = getattr(text, "torch_dtype", None) or getattr(text, "dtype", None)
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "float16": torch.float16,
              "fp16": torch.float16, "float32": torch.float32}.get(dt.lower(), getattr(torch, dt, None))
    return dt or torch.bfloat16


# token / sequence buckets of the captured PREFILL steps (capture_prefill_graphs)
# (steps of ~1/8 above 256 tokens: a step is padded to its bucket, and these steps are device-bound)
PREFILL_GRAPH_TOKENS = (64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048, 2560, 3072, 3584, 4096)
PREFILL_GRAPH_SEQS = (1, 2, 4)
# Steps beyond that table (a full house: 16 x 1024 tokens) get a graph LAZILY, keyed by (tokens up to a multiple of 256,
# sequences up to a power of two, longest query up to a power of two >= 256): captured when a key comes by the second
# time (or is announced: ensure_prefill_graph), at most PREFILL_LAZY_GRAPHS of them alive, the least recently replayed
# one evicted; a step that would be padded by more than 1/16 of its tokens stays eager (these steps are device-bound).
PREFILL_LAZY_GRAPHS = 8
PREFILL_LAZY_SEQS_CAP = 64  # rows of the static metadata buffer's block table (it is uploaded whole with every step)
# Tensor parallelism: a captured prefill step all-reduces its activations over the xGMI exchange region (RCCL on
# prefill-sized tensors stays outside graphs), whose slots are 2 x world x tokens x hidden x 2 bytes per rank: the table
# stops here (Qwen3-32B at TP 8: 335 MB per rank)
TP_PREFILL_GRAPH_TOKENS = 2048
# Stream capture in THREAD-LOCAL error mode: with a process group alive, ProcessGroupNCCL's watchdog thread polls the
# events of enqueued collectives (hipEventQuery); under the default global mode such a call from ANOTHER thread while
# this one captures is "operation not permitted when stream is capturing" - raised inside the watchdog, which terminates
# the process.  Only this thread's own calls need policing during a capture.
CAPTURE_MODE = "thread_local"


def graph_buckets(max_num_seqs: int) -> list[int]:
    b, out = 1, []
    while b < max_num_seqs:
        out.append(b)
        b *= 2
    out.append(max_num_seqs)
    return out


class ModelRunner:
    def __init__(self, config: Config, rank: int, event=None):
        self.config = config
        self.hf_config = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.event = event

        if not torch.cuda.is_available():
            raise RuntimeError("nanovllm (MI355X build) needs a HIP device: there is no CPU execution path")
        local = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(self.device)
        from nanovllm.layers import parallel

        # MI355_TP1_COLLECTIVES=1 (layers/parallel.py): a one-rank engine that takes every RCCL code path of a
        # tensor-parallel rank - the bring-up check of the "nccl" branches on a single GPU
        self.collective = self.world_size > 1 or os.environ.get("MI355_TP1_COLLECTIVES", "0") == "1"
        self._own_group = False
        if self.collective and not dist.is_initialized():
            # backend "nccl" is RCCL on ROCm; rendezvous on the loopback interface.
            # MI355_DIST_BACKEND=gloo lets several ranks share one GPU (functional TP tests on a
            # 1-GPU box; not capturable into graphs).
            backend = os.environ.get("MI355_DIST_BACKEND", "nccl")
            kw = {"device_id": self.device} if backend == "nccl" else {}
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{config.hccl_port}",
                                    world_size=self.world_size, rank=rank, **kw)
            self._own_group = True
        parallel.set_force_collectives(self.collective and self.world_size == 1)

        # the TP group is what Config says, not whatever process group happens to exist: independent replicas
        # (bench.py --mode replicas) share a default group only for their barrier
        parallel.set_tp(rank if self.world_size > 1 else 0, self.world_size)
        parallel.set_xgmi_comm(None)
        self.channel = None
        if self.world_size > 1:
            from nanovllm.engine.rpc import slot_words

            self.channel = StepChannel(config.hccl_port, self.world_size, rank,
                                       slot_words(config.max_num_batched_tokens, config.max_num_seqs,
                                                  config.max_model_len, config.kvcache_block_size))
            self.channel.skip_cached_prefix = bool(config.prefix_aware_prefill)
        self.xgmi = None
        self._steps_run = 0
        if self.world_size > 1:
            from nanovllm.layers import xgmi_comm

            rows = max(64, min(config.max_num_seqs, 512))  # decode-sized activations; eager prefill goes through RCCL
            if config.use_graphs and config.prefill_graphs and os.environ.get("MI355_PREFILL_GRAPHS", "1") != "0":
                # ... captured prefill steps (capture_prefill_graphs) through the region as well: every rank replays the
                # same graph, whose all-reduces must be kernels of this library, not RCCL calls
                rows = max(rows, max([t for t in PREFILL_GRAPH_TOKENS
                                      if t <= min(config.max_num_batched_tokens, TP_PREFILL_GRAPH_TOKENS)] or [0]))
            eager_rows = max(64, min(config.max_num_seqs, 512))
            self.xgmi = xgmi_comm.create_if_enabled(rank, self.world_size, rows * self.hf_config.hidden_size * 2,
                                                    self.device)
            if self.xgmi is not None and os.environ.get("MI355_XGMI_EAGER_LARGE", "0") != "1":
                # eager launches: decode-sized rows only (XgmiComm.eager_max_bytes; the switch: bring-up / repro)
                self.xgmi.eager_max_bytes = eager_rows * self.hf_config.hidden_size * 2
            parallel.set_xgmi_comm(self.xgmi)
            self.xgmi_selftest = xgmi_comm.LAST_STATUS

        dtype = _torch_dtype_of(self.hf_config)
        if dtype != torch.bfloat16:
            raise NotImplementedError(f"the gfx950 kernels are bf16; checkpoint dtype is {dtype}")
        self._check_supported_shapes()
        from nanovllm.layers.linear import set_weight_quantization

        set_weight_quantization(config.quantization)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device(self.device):
                arch = self.hf_config.architectures[0] if getattr(self.hf_config, "architectures", None) \
                    else "Qwen3ForCausalLM"
                self.model = model_dict[arch](self.hf_config)
        finally:
            torch.set_default_dtype(prev)
        from nanovllm.layers.linear import LinearBase, check_linear_shape

        for name, module in self.model.named_modules():  # fail at start-up, not in the middle of a prefill step
            if isinstance(module, LinearBase):
                check_linear_shape(name, *module.weight.shape)
        if has_checkpoint(config.model):
            load_model(self.model, config.model)
            self.synthetic = False
        else:
            init_synthetic_weights(self.model, self.hf_config, seed=config.synthetic_seed)
            self.synthetic = True
        self.model.eval()
        self.sampler = Sampler(seed=config.sampling_seed)
        if self.collective:
            # every rank draws the sampler's noise for its vocabulary shard: one seed (rank 0's; a seed of None is
            # drawn from os.urandom per process) and one step counter, advanced in lockstep (run / launch_decode)
            seed = torch.tensor([self.sampler.seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64,
                                device=self.device if dist.get_backend() == "nccl" else "cpu")
            parallel.STATS["rccl_broadcast"] += dist.get_backend() == "nccl"
            dist.broadcast(seed, 0)
            self.sampler.seed = int(seed.item())
        # parity hook under tensor parallelism: gather the full logits to rank 0 although the graph already holds
        # the step's tokens (tests compare them with the one-GPU run); a collective, so every rank reads the switch
        self.gather_logits = os.environ.get("MI355_TP_GATHER_LOGITS", "0") != "0"
        self._xgmi_flag = torch.zeros(1, dtype=torch.int32).pin_memory() if self.world_size > 1 else None
        torch.cuda.empty_cache()
        self.allocate_kv_cache()
        self._alloc_staging()
        self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
        self.graph_samples: set[int] = set()  # buckets whose graph ends in the token choice, not in the logits
        self.graph_logits: dict[int, torch.Tensor] = {}
        # with TP the captured graph holds the xGMI exchange kernels; without them (RCCL all-reduce inside)
        # capture is only attempted on the nccl backend
        if self.collective:
            dist.barrier()  # weight loading can skew the ranks by more than the exchange kernels' patience
        if config.use_graphs and (not self.collective or self.xgmi is not None or dist.get_backend() == "nccl"):
            try:
                self.capture_decode_graphs()
            except Exception as e:  # e.g. a collective that refuses stream capture: run eagerly instead
                import warnings

                warnings.warn(f"hipGraph capture failed ({e!r}); decoding eagerly")
                self.graphs.clear()
                self.graph_logits.clear()
                reset_context()
        self.prefill_graphs: dict[tuple, torch.cuda.CUDAGraph] = {}  # (tb, sb): start-up table; (tb, sb, mq): lazy
        self.prefill_graph_logits: dict[tuple, torch.Tensor] = {}
        self.prefill_graph_replays = 0
        self.prefill_graph_lazy_captures = 0
        self._pg_pool = None
        self._pg_lazy_seen: dict[tuple, int] = {}   # sightings of keys without a graph (bounded, see _prefill_bucket)
        self._pg_lazy_lru: list[tuple] = []         # lazily captured keys, least recently replayed first
        self._pg_lazy_on = False
        # one GPU, or tensor-parallel ranks whose exchange region is up (then every collective of a step is a kernel of
        # this library and the token choice is made among the ranks inside the graph, as in the decode graphs)
        capturable = self.can_launch_prefill or (self.world_size > 1 and self.xgmi is not None)
        if (config.use_graphs and config.prefill_graphs and self.graph_samples and capturable
                and os.environ.get("MI355_PREFILL_GRAPHS", "1") != "0"):
            ok = True
            try:
                if self.world_size > 1:
                    dist.barrier()  # (the capture's warm-up runs exchange with the peers: start together)
                self.capture_prefill_graphs()
                self._pg_lazy_on = (bool(self.prefill_graphs) and self.world_size == 1
                                    and os.environ.get("MI355_PREFILL_GRAPHS_LAZY", "1") != "0")
            except Exception as e:
                import warnings

                warnings.warn(f"hipGraph capture of the prefill buckets failed ({e!r}); prefill steps are launched eagerly")
                ok = False
            if self.world_size > 1:
                # every rank replays the same graph or none does: agree (a rank that failed mid-capture has left the
                # exchange epochs uneven, which the status check of the next step would report - it does not hang)
                flag = torch.tensor([int(ok)], dtype=torch.int32,
                                    device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item())
            if not ok:
                self.prefill_graphs.clear()
                self.prefill_graph_logits.clear()
                self._pg_lazy_on = False
                reset_context()
        if self.collective:
            dist.barrier()

    def _check_supported_shapes(self):
        """What the attention kernels are compiled for, checked before the model is built and before any
        graph capture (a failure there would surface as a bare MI_EUNSUPPORTED inside warm-up)."""
        hf = getattr(self.hf_config, "text_config", self.hf_config)
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        hq, hkv = hf.num_attention_heads, hf.num_key_value_heads
        from nanovllm import ops

        if hq % hkv:
            raise NotImplementedError(f"{hq} query heads over {hkv} kv heads: not a whole GQA group size")
        if ops.attention_is_plain(hq, hkv, head_dim) and not ops.attention_plain_supported(hq, hkv, head_dim):
            raise NotImplementedError(
                f"head_dim {head_dim} with {hq} query heads over {hkv} kv heads: the fragment-native attention kernels "
                "take head_dim 64 or 128 with GQA groups 1, 2, 4, 7, 8, 16; the plain-layout kernels head_dim 64 or 128 "
                "with any group up to 8")
        if hq % self.world_size or hkv % self.world_size:
            raise ValueError(f"tensor_parallel_size {self.world_size} does not divide {hq} query / {hkv} kv heads")
        if getattr(hf, "num_experts", 0) and getattr(hf, "moe_intermediate_size", 0):
            from nanovllm._C import lib

            inter = hf.moe_intermediate_size
            if inter % self.world_size:
                raise ValueError(f"tensor_parallel_size {self.world_size} does not divide moe_intermediate_size {inter}")
            if lib.mi_moe_shapes_supported(hf.hidden_size, inter // self.world_size) != 0:
                raise NotImplementedError(
                    f"MoE experts of hidden {hf.hidden_size} x intermediate {inter // self.world_size} per rank: the "
                    "grouped expert GEMMs (csrc/moe.hip) have no instantiation for these contraction lengths")

    # ------------------------------------------------------------------ lifecycle / RPC
    def exit(self, abort: bool = False):
        """abort: a peer is known not to reach the exit barrier (rank 0 raised after an exchange time-out)."""
        from nanovllm.layers import parallel

        parallel.reset_tp()
        parallel.set_force_collectives(False)
        self.graphs.clear()
        self.graph_logits.clear()
        self.prefill_graphs.clear()
        self.prefill_graph_logits.clear()
        torch.cuda.synchronize()
        if self.channel is not None:
            if not abort:
                dist.barrier()
            self.channel.close()
        if self.xgmi is not None:
            from nanovllm.layers import parallel

            try:
                self._check_xgmi()
            finally:
                parallel.set_xgmi_comm(None)
            self.xgmi.close()
            self.xgmi = None
        if self.collective and dist.is_initialized() and (self.world_size > 1 or self._own_group):
            dist.destroy_process_group()

    def loop(self):
        """TP worker main loop (rank > 0): execute whatever rank 0 publishes."""
        from nanovllm.engine import host_gc
        from nanovllm.engine.rpc import ChannelError

        frozen = self.config.gc_control and os.environ.get("MI355_GC_CONTROL", "1") != "0"
        if frozen:
            # a full collection on ONE worker stalls every rank at the next exchange for as long as it takes (~100 ms with
            # torch + the model alive); everything alive now is permanent: out of the collector's sight (engine/host_gc.py)
            host_gc.freeze_permanent_heap()

        def leave(abort: bool):
            stats = os.environ.get("MI355_WORKER_STATS")  # tests / bring-up: what this worker ran, as <prefix>.<rank>.json
            if stats:
                import json

                with open(f"{stats}.{self.rank}.json", "w") as f:
                    json.dump({"rank": self.rank, "steps_run": self._steps_run,
                               "prefill_graph_replays": self.prefill_graph_replays,
                               "prefill_graphs": len(self.prefill_graphs),
                               "lookahead_launches": getattr(self, "lookahead_launches", 0)}, f)
            self.exit(abort=abort)
            if frozen:
                host_gc.release_permanent_heap()

        while True:
            try:
                method, seqs, is_prefill, extra = self.channel.recv()
            except ChannelError:  # a torn message: never replay it - leave without the exit barrier, loudly
                leave(True)
                raise
            if method == "exit":
                leave(False)
                return
            if method == "abort":  # rank 0 gave up on a step (an exchange timed out): leave without the exit barrier
                leave(True)
                return
            if method == "launch_decode":  # a step rank 0 queued behind the running one: queue the same step here
                self.launch_decode(seqs, extra if extra else None)
            elif method == "launch_prefill":  # a captured prefill step: stage the same metadata, replay the same graph
                self.launch_prefill(seqs)
            else:
                self.run(seqs, is_prefill)

    def call(self, method_name: str, *args):
        if method_name == "run" and not args[0]:
            return []  # nothing scheduled (everything preempted): no message, no collective, on any rank
        if self.channel is not None and self.rank == 0:
            if method_name == "launch_decode":
                self.channel.send(method_name, args[0], False, extra=args[1] if len(args) > 1 else None)
            elif method_name == "launch_prefill":  # (the bucket is every rank's own, equal, decision)
                self.channel.send(method_name, args[0], True)
            else:
                self.channel.send(method_name, *args)
        return getattr(self, method_name)(*args)

    # ------------------------------------------------------------------ KV cache
    def allocate_kv_cache(self):
        """model_runner.py:195-229: size from free memory, ONE tensor [2, L, ...], per-layer
        slices bound to every module that has k_cache/v_cache.  Layout per layer is the
        fragment-native [nblk, Hkv, block/16, 2048] of include/mi355_nanovllm.h."""
        cfg, hf = self.config, getattr(self.hf_config, "text_config", self.hf_config)
        free, total = torch.cuda.mem_get_info(self.device)
        used = total - free
        stats = torch.cuda.memory_stats(self.device)
        peak = stats.get("allocated_bytes.all.peak", 0)
        current = stats.get("allocated_bytes.all.current", 0)
        n_kv = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        layers = hf.num_hidden_layers
        block_bytes = 2 * layers * self.block_size * n_kv * head_dim * 2
        if cfg.num_kvcache_blocks <= 0:
            available = total * cfg.gpu_memory_utilization - used - peak + current
            cfg.num_kvcache_blocks = int(available) // block_bytes
            if self.collective:
                # rank 0's scheduler hands out block ids to every rank: all ranks must allocate the same
                # number of blocks, i.e. what the rank with the least free memory can hold (the reference
                # sizes per rank, model_runner.py:195-214, and would index past the smaller caches)
                n = torch.tensor([cfg.num_kvcache_blocks], dtype=torch.int64,
                                 device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(n, op=dist.ReduceOp.MIN)
                cfg.num_kvcache_blocks = int(n.item())
        assert cfg.num_kvcache_blocks > 0, "no memory left for even one KV cache block"
        # an odd number of allocated blocks keeps the per-layer stride (and with it the distance between
        # a tile's K and V copies, which one wavefront loads together) off large powers of two: with
        # 4096 blocks K and V of a tile shared all low 27 address bits and the attention kernel lost ~4 %
        alloc_blocks = cfg.num_kvcache_blocks | 1
        from nanovllm import ops

        hq = hf.num_attention_heads // self.world_size
        if ops.attention_is_plain(hq, n_kv, head_dim):  # head_dim 64 / odd GQA groups: [blocks, kv heads, block, head_dim]
            shape = ops.kv_cache_shape_plain(alloc_blocks, n_kv, self.block_size, head_dim)
        else:
            shape = ops.kv_cache_shape(alloc_blocks, n_kv, self.block_size, head_dim)
        self.kv_cache = torch.zeros((2, layers, *shape), dtype=torch.bfloat16, device=self.device)
        layer_id = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer_id, : cfg.num_kvcache_blocks]
                module.v_cache = self.kv_cache[1, layer_id, : cfg.num_kvcache_blocks]
                layer_id += 1
        assert layer_id == layers

    # ------------------------------------------------------------------ staging buffers
    def _alloc_staging(self):
        cfg = self.config
        B = cfg.max_num_seqs
        W = -(-(cfg.max_model_len + 1) // self.block_size)  # a max-length prompt still decodes one token
        self.table_cols = W
        # [ids i64 B][pos i64 B][rng u64 2][temps f32 B][ctx i32 B][src i32 B][slots i32 2B][tables i32 B*W]
        off, spans = 0, {}
        for name, nbytes in (("ids", 8 * B), ("pos", 8 * B), ("rng", 16), ("temps", 4 * B), ("ctx", 4 * B),
                             ("src", 4 * B), ("slots", 8 * B), ("tables", 4 * B * W)):
            spans[name] = (off, nbytes)
            off += (nbytes + 15) // 16 * 16
        self._stage_bytes = off
        self._stage_kernel = os.environ.get("MI355_STAGE_KERNEL", "1") != "0"
        self.dev_stage = torch.zeros(off, dtype=torch.uint8, device=self.device)

        def views(buf):
            def v(name, dtype, shape):
                o, n = spans[name]
                return buf[o:o + n].view(dtype).view(shape)
            return {"ids": v("ids", torch.int64, (B,)), "pos": v("pos", torch.int64, (B,)),
                    "rng": v("rng", torch.int64, (2,)), "temps": v("temps", torch.float32, (B,)),
                    "ctx": v("ctx", torch.int32, (B,)), "src": v("src", torch.int32, (B,)),
                    "slots": v("slots", torch.int32, (B, 2)), "tables": v("tables", torch.int32, (B, W))}

        self.dev = views(self.dev_stage)
        # TWO pinned staging buffers (+ token landing buffers and events), used alternately: a step queued
        # behind the one still running (launch_decode) must not overwrite metadata whose upload has not run yet
        self.host_stages, self.hosts, self.stagers = [], [], []
        for _ in range(2):
            buf = torch.empty(off, dtype=torch.uint8, pin_memory=True)
            h = {k: t.numpy() for k, t in views(buf).items()}
            h["src"][:] = -1
            self.host_stages.append(buf)
            self.hosts.append(h)
            self.stagers.append(batch_meta.DecodeStager(h["ids"], h["pos"], h["ctx"], h["slots"], h["tables"],
                                                        h["temps"]))
        self._flip = 0
        self._src_dirty = [False, False]
        self.tokens_dev = torch.zeros(B, dtype=torch.int64, device=self.device)
        self.tokens_hosts = [torch.zeros(B, dtype=torch.int64, pin_memory=True) for _ in range(2)]
        self.step_events = [torch.cuda.Event() for _ in range(2)]
        self._events_recorded = [False, False]
        # prefill metadata staging: ids + positions (8 B) + slots (4 B) per token, per-sequence vectors, tables
        nbytes = cfg.max_num_batched_tokens * 20 + B * (W + 5) * 4 + 4096
        # (two pinned buffers, alternating: launch_prefill queues a step behind the one whose upload may not have run;
        # ONE device buffer - the uploads are ordered on the stream behind the kernels that read the previous contents)
        self.prefill_hosts = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self.prefill_dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._pflip = 0
        self.prefill_tokens_hosts = [torch.zeros(B, dtype=torch.int64, pin_memory=True) for _ in range(2)]
        # (start, end) of a queued prefill step on the device: the end event is what collect_prefill waits for, the pair
        # gives the step's device time to the engine's prefill trace
        self.prefill_starts = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.prefill_events = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self._prefill_events_recorded = [False, False]
        # captured prefill steps (capture_prefill_graphs) read their metadata at FIXED addresses: one static device buffer
        # [ids i64 T][pos i64 T][slots i32 T][cu_q i32 S+1][cu_k i32 S+1][kv_lens i32 S][temps f32 S][rng u64 2][tables i32 S*W]
        # for the largest bucket, of which a graph uses the leading T_b / S_b entries; two pinned mirrors, alternating
        t_cap = cfg.max_num_batched_tokens if self.world_size == 1 else min(cfg.max_num_batched_tokens,
                                                                            TP_PREFILL_GRAPH_TOKENS)
        self._pg_tmax = max([t for t in PREFILL_GRAPH_TOKENS if t <= t_cap] or [0])
        self._pg_smax = max([n for n in PREFILL_GRAPH_SEQS if n <= cfg.max_num_seqs] or [0])
        # (the buffer itself holds the largest step there is: the lazily captured large steps read it too)
        tmax = (cfg.max_num_batched_tokens if self.world_size == 1 else self._pg_tmax) if self._pg_tmax else 0
        smax = (min(cfg.max_num_seqs, PREFILL_LAZY_SEQS_CAP) if self.world_size == 1 else self._pg_smax) if self._pg_smax else 0
        self._pg_tcap, self._pg_scap = tmax, smax
        if tmax and smax:
            off, sp = 0, {}
            for name, nb in (("ids", 8 * tmax), ("pos", 8 * tmax), ("slots", 4 * tmax), ("cu_q", 4 * (smax + 1)),
                             ("cu_k", 4 * (smax + 1)), ("kv_lens", 4 * smax), ("temps", 4 * smax), ("rng", 16),
                             ("tables", 4 * smax * W)):
                sp[name] = (off, nb)
                off += (nb + 15) // 16 * 16

            def pviews(buf):
                def v(name, dtype, shape):
                    o, n = sp[name]
                    return buf[o:o + n].view(dtype).view(shape)
                return {"ids": v("ids", torch.int64, (tmax,)), "pos": v("pos", torch.int64, (tmax,)),
                        "slots": v("slots", torch.int32, (tmax,)), "cu_q": v("cu_q", torch.int32, (smax + 1,)),
                        "cu_k": v("cu_k", torch.int32, (smax + 1,)), "kv_lens": v("kv_lens", torch.int32, (smax,)),
                        "temps": v("temps", torch.float32, (smax,)), "rng": v("rng", torch.int64, (2,)),
                        "tables": v("tables", torch.int32, (smax, W))}

            self.pg_dev_buf = torch.zeros(off, dtype=torch.uint8, device=self.device)
            self.pg_dev = pviews(self.pg_dev_buf)
            self.pg_host_bufs = [torch.zeros(off, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self.pg_hosts = [{k: t.numpy() for k, t in pviews(b).items()} for b in self.pg_host_bufs]

    def _upload(self, dev: torch.Tensor, pinned: torch.Tensor) -> None:
        """A step's staged metadata: pinned host buffer -> its device buffer, queued on the step's stream.  As a kernel
        (ops.stage_copy) where the sizes allow: torch's async copy of a few tens of KB goes through the copy engine and
        leaves ~25 us of idle device between two decode graphs (MI355_STAGE_KERNEL=0: the async copy)."""
        if self._stage_kernel and dev.numel() % 16 == 0 and dev.numel() == pinned.numel():
            ops.stage_copy(dev, pinned)
        else:
            dev.copy_(pinned, non_blocking=True)

    def _download_tokens(self, landing: torch.Tensor, real: int) -> None:
        """The step's sampled tokens -> a pinned landing buffer, queued behind the step (the same kernel: the whole
        token buffer, a few hundred bytes; rows beyond `real` are padding)."""
        if self._stage_kernel and landing.numel() == self.tokens_dev.numel() and landing.numel() % 2 == 0:
            ops.stage_copy(landing, self.tokens_dev)
        else:
            landing[:real].copy_(self.tokens_dev[:real], non_blocking=True)

    def _fill_decode_stage(self, seqs: list[Sequence], bucket: int, src_rows=None) -> int:
        """decode_meta(seqs, pad_to=bucket, dummy slot in the reserved last block) into the next pinned
        staging buffer (incremental block-table rows), then ONE async copy to the device.  src_rows[i] >= 0:
        row i's input id is the token the previous step sampled in that row (still on the device).
        Returns the index of the staging buffer used."""
        self._flip ^= 1
        b = self._flip
        self.stagers[b].fill(seqs, bucket, self.config.num_kvcache_blocks - 1)
        h = self.hosts[b]
        rng = h["rng"].view(np.uint64)  # what THIS step's sampler uses (a captured graph reads it here)
        rng[0] = self.sampler.seed & 0xFFFFFFFFFFFFFFFF
        rng[1] = (self.sampler.step + 1) & 0xFFFFFFFFFFFFFFFF
        if src_rows is not None:
            h["src"][:len(src_rows)] = src_rows
            h["src"][len(src_rows):bucket] = -1
            self._src_dirty[b] = True
        elif self._src_dirty[b]:
            h["src"][:] = -1
            self._src_dirty[b] = False
        self._upload(self.dev_stage, self.host_stages[b])
        return b

    # ------------------------------------------------------------------ metadata -> context
    def prepare_prefill(self, seqs: list[Sequence]):
        """prefill_meta() packed into ONE pinned staging buffer and uploaded with ONE async copy (the
        reference issues seven pinned allocations + copies, model_runner.py:271-290).  The two pinned buffers
        alternate: the upload of the step before last is done (its tokens were collected) when one is reused."""
        m = batch_meta.prefill_meta(seqs, self.block_size, skip_cached=self.config.prefix_aware_prefill)
        self._pflip ^= 1
        # (the sampler's temperatures travel in the same staging buffer: a pinned allocation per step - the reference's
        # prepare_sample, model_runner.py:368-372 - is a first-use cost in every new size class, i.e. in a request's TTFT)
        temps = np.fromiter((0.0 if s.greedy else s.temperature for s in seqs), dtype=np.float32, count=len(seqs))
        arrays = (m.input_ids, m.positions, m.slot_mapping, m.cu_seqlens_q, m.cu_seqlens_k, m.kv_lens,
                  np.ascontiguousarray(m.block_tables), temps)
        offs, off = [], 0
        for a in arrays:
            offs.append(off)
            off += (a.nbytes + 15) // 16 * 16
        if off > self.prefill_dev.numel():  # grow (rare: sized for max_num_batched_tokens up front)
            self.prefill_dev = torch.empty(off * 2, dtype=torch.uint8, device=self.device)
        if off > self.prefill_hosts[self._pflip].numel():
            self.prefill_hosts[self._pflip] = torch.empty(off * 2, dtype=torch.uint8, pin_memory=True)
        pinned = self.prefill_hosts[self._pflip]
        host = pinned.numpy()
        for a, o in zip(arrays, offs):
            host[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        self.prefill_dev[:off].copy_(pinned[:off], non_blocking=True)

        def dev(i, dtype, shape=None):
            a = arrays[i]
            t = self.prefill_dev[offs[i]:offs[i] + a.nbytes].view(dtype)
            return t if shape is None else t.view(shape)

        set_context(True, cu_seqlens_q=dev(3, torch.int32), cu_seqlens_k=dev(4, torch.int32),
                    max_seqlen_q=m.max_seqlen_q, max_seqlen_k=m.max_seqlen_k, slot_mapping=dev(2, torch.int32),
                    context_lens=None, block_tables=dev(6, torch.int32, tuple(m.block_tables.shape)),
                    block_size=self.block_size, kv_lens=dev(5, torch.int32))
        self._prefill_temps = dev(7, torch.float32)
        return dev(0, torch.int64), dev(1, torch.int64)

    def prepare_decode(self, seqs: list[Sequence], bucket: int | None = None):
        """Eager decode uses the same staging buffers with bucket == real batch."""
        real = len(seqs)
        bucket = bucket or real
        self._fill_decode_stage(seqs, bucket)
        d = self.dev
        set_context(False, slot_mapping=d["slots"][:bucket], context_lens=d["ctx"][:bucket],
                    block_tables=d["tables"][:bucket], is_enforce_eager=not self.config.use_graphs,
                    real_bs=real, block_size=self.block_size)
        return d["ids"][:bucket], d["pos"][:bucket]

    def prepare_sample(self, seqs: list[Sequence]):
        """The temperatures of the prefill step just staged (prepare_prefill uploaded them with the step's metadata)."""
        assert self._prefill_temps.numel() == len(seqs)
        return self._prefill_temps

    # ------------------------------------------------------------------ graphs
    @torch.inference_mode()
    def capture_decode_graphs(self):
        cfg, d = self.config, self.dev
        # the token choice is part of the graph on one GPU and, with the exchange region up, under tensor parallelism
        # (every rank then holds the step's tokens: no logits gather, no sampler on rank 0, no host in the loop)
        pick = os.environ.get("MI355_GRAPH_SAMPLER", "1") != "0" and self.model.lm_head.can_pick()
        self.graph_samples = {bs for bs in graph_buckets(cfg.max_num_seqs) if pick and bs <= ops.SKINNY_MAX_M}
        # under TP a sampling graph ends in the candidate exchange over the xGMI region: that collective is what keeps
        # the ranks' queued steps (engine lookahead) in lockstep
        assert not (self.world_size > 1 and self.graph_samples) or self.xgmi is not None
        # neutral metadata: every row padded (context_len 0, dummy slot)
        self._fill_decode_stage([], cfg.max_num_seqs)
        pool = None
        for bs in reversed(graph_buckets(cfg.max_num_seqs)):
            set_context(False, slot_mapping=d["slots"][:bs], context_lens=d["ctx"][:bs],
                        block_tables=d["tables"][:bs], is_enforce_eager=False, real_bs=bs,
                        block_size=self.block_size,
                        token_src=d["src"][:bs] if bs in self.graph_samples else None,
                        prev_tokens=self.tokens_dev if bs in self.graph_samples else None)

            def body(bs=bs):
                hidden = self.model(d["ids"][:bs], d["pos"][:bs])
                if bs in self.graph_samples:  # one GPU: the head GEMM picks the tokens as it writes the logits
                    return self.model.lm_head.local_logits_pick(hidden, d["temps"][:bs], d["rng"],
                                                                self.tokens_dev[:bs])
                return self.model.lm_head.local_logits(hidden)

            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up run outside capture (workspaces, exchange epochs)
                body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode=CAPTURE_MODE):  # kernels only: the logits gather of TP stays outside
                logits = body()
            pool = pool or graph.pool()
            self.graphs[bs] = graph
            self.graph_logits[bs] = logits
        reset_context()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ captured prefill steps
    @torch.inference_mode()
    def capture_prefill_graphs(self):
        """Prefill steps of up to PREFILL_GRAPH_SEQS sequences and PREFILL_GRAPH_TOKENS tokens as hipGraphs, one per
        (token bucket, sequence bucket) - what a serving engine runs for almost every arrival.  An eager prefill step is
        ~250 launches = 3.6 ms of host time whatever its size, which for a 500-token prompt is four times its device time
        and most of an open-loop request's TTFT; a replay costs the host 0.1 ms.  (The reference keeps prefill eager:
        model_runner.py:393 bypasses its compiled graph for everything but decode.)
        A step is padded to its bucket: pad tokens are id 0 at position 0 with slot -1 (the store kernels skip them), pad
        sequences have length 0 (the attention kernel's workgroups for them exit on their query length), so the real rows
        are computed exactly as in an eager step of T_b rows; the step's last-token rows go through the head GEMM with the
        pick epilogue (the sampler's keys, {seed, step} read from the static buffer), as the decode graphs do."""
        cfg = self.config
        text = getattr(self.hf_config, "text_config", self.hf_config)
        if not self._pg_tmax or getattr(text, "num_experts", 0):
            return  # (the sparse block's routing over pad tokens is not worth a graph: eager)
        for tb in sorted((t for t in PREFILL_GRAPH_TOKENS if t <= self._pg_tmax), reverse=True):
            for sb in sorted((n for n in PREFILL_GRAPH_SEQS if n <= self._pg_smax and n <= tb), reverse=True):
                if -(-tb // sb) > cfg.max_model_len:
                    continue  # no step of sb sequences has that many tokens (a sequence is at most max_model_len long)
                self._capture_prefill_graph((tb, sb))
        reset_context()
        torch.cuda.synchronize()

    def _capture_prefill_graph(self, key: tuple):
        """One captured prefill step: key (tb, sb) of the start-up table (the attention grid covers queries of up to tb
        tokens) or (tb, sb, mq) of a lazily captured large step (queries of up to mq tokens).  Leaves the alternation
        of the pinned staging buffers as it found it (a queued step's handle names its buffer)."""
        tb, sb = key[0], key[1]
        mq = key[2] if len(key) > 2 else tb
        d, flip = self.pg_dev, self._pflip
        self._stage_prefill_static([], tb, sb)
        set_context(True, cu_seqlens_q=d["cu_q"][:sb + 1], cu_seqlens_k=d["cu_k"][:sb + 1], max_seqlen_q=mq,
                    max_seqlen_k=mq, slot_mapping=d["slots"][:tb], block_tables=d["tables"][:sb],
                    block_size=self.block_size, kv_lens=d["kv_lens"][:sb])

        head = self.model.lm_head

        def body():
            hidden = self.model(d["ids"][:tb], d["pos"][:tb])
            x = ops.gather_last_tokens(hidden, d["cu_q"][:sb + 1])
            if self.world_size == 1:
                logits, _ = ops.gemm_packed_pick(x, head.weight_packed, d["temps"][:sb], d["rng"], self.tokens_dev[:sb])
                return logits
            # tensor parallelism: every rank picks in its vocabulary shard, the ranks exchange {key, token} pairs over the
            # region and take the same winner (ParallelLMHead.local_logits_pick, the decode graphs' last launches)
            pairs = torch.empty((sb, 2), dtype=torch.int32, device=x.device)
            logits, _ = ops.gemm_packed_pick(x, head.weight_packed, d["temps"][:sb], d["rng"], self.tokens_dev[:sb],
                                             col_offset=head.vocab_start_idx, pairs_out=pairs)
            self.xgmi.pick_exchange(pairs, self.tokens_dev[:sb])
            return logits

        import contextlib

        # (a captured step's ranks replay together: its prefill-sized all-reduces are exchange kernels, XgmiComm.large)
        whole_region = self.xgmi.large() if self.xgmi is not None else contextlib.nullcontext()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with whole_region, torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        whole_region = self.xgmi.large() if self.xgmi is not None else contextlib.nullcontext()
        with whole_region, torch.cuda.graph(graph, pool=self._pg_pool, capture_error_mode=CAPTURE_MODE):
            logits = body()
        self._pg_pool = self._pg_pool or graph.pool()
        self.prefill_graphs[key] = graph
        self.prefill_graph_logits[key] = logits
        self._pflip = flip

    @staticmethod
    def _lazy_prefill_key(tokens: int, n_seqs: int, max_q: int) -> tuple[int, int, int] | None:
        tb = -(-tokens // 256) * 256
        if (tb - tokens) * 16 > tokens:
            return None  # padded by more than 1/16: a device-bound step is better off eager
        sb = 1 << max(0, n_seqs - 1).bit_length()
        mq = max(256, 1 << max(0, max_q - 1).bit_length())
        return (tb, sb, mq)

    @torch.inference_mode()
    def ensure_prefill_graph(self, tokens: int, n_seqs: int, max_q: int) -> bool:
        """Capture the large-step graph of this shape now (the warm-up announces the full-house shape with it)."""
        key = self._lazy_prefill_key(tokens, n_seqs, max_q) if self._pg_lazy_on else None
        if key is None or key[0] > self._pg_tcap or key[1] > self._pg_scap:
            return False
        if key not in self.prefill_graphs:
            self._capture_lazy(key)
        return True

    def _capture_lazy(self, key: tuple):
        if len(self._pg_lazy_lru) >= PREFILL_LAZY_GRAPHS:
            torch.cuda.synchronize()  # (a queued step may still be replaying the graph that is about to be destroyed)
        while len(self._pg_lazy_lru) >= PREFILL_LAZY_GRAPHS:
            old = self._pg_lazy_lru.pop(0)
            self.prefill_graphs.pop(old, None)
            self.prefill_graph_logits.pop(old, None)
        self._capture_prefill_graph(key)
        reset_context()
        torch.cuda.synchronize()
        self._pg_lazy_lru.append(key)
        self._pg_lazy_seen.pop(key, None)
        self.prefill_graph_lazy_captures += 1

    def prefill_graph_takes(self, n_seqs: int, n_tokens: int) -> bool:
        """Is a prefill step of n_seqs sequences and at most n_tokens tokens a graph replay (engine: may it be queued
        behind a running decode step without holding that step's tokens back by an eager launch sequence)?"""
        if not self.prefill_graphs or n_seqs > self._pg_smax or n_tokens > self._pg_tmax:
            return False
        tb = next(t for t in PREFILL_GRAPH_TOKENS if t >= n_tokens)
        sb = next(n for n in PREFILL_GRAPH_SEQS if n >= n_seqs)
        return (tb, sb) in self.prefill_graphs

    def _prefill_bucket(self, seqs: list[Sequence], note: bool = False) -> tuple | None:
        """The captured step this prefill step replays, or None (eager).  note=True (once per step, by launch_prefill /
        run): a large step without a graph is counted, and captured when its key has come by before."""
        if not self.prefill_graphs or len(seqs) > self._pg_scap:
            return None
        skip = self.config.prefix_aware_prefill
        q_lens = [len(s) - (min(s.num_prefix_tokens, len(s) - 1) if skip else 0) for s in seqs]
        tokens = sum(q_lens)
        if tokens > self._pg_tcap or max(len(s.block_table) for s in seqs) > self.table_cols:
            return None
        if tokens <= self._pg_tmax and len(seqs) <= self._pg_smax:
            tb = next(t for t in PREFILL_GRAPH_TOKENS if t >= tokens)
            sb = next(n for n in PREFILL_GRAPH_SEQS if n >= len(seqs))
            if (tb, sb) in self.prefill_graphs:
                return (tb, sb)
        if not self._pg_lazy_on:
            return None
        key = self._lazy_prefill_key(tokens, len(seqs), max(q_lens))
        if key is None or key[0] > self._pg_tcap or key[1] > self._pg_scap:
            return None
        if key in self.prefill_graphs:
            if note:
                self._pg_lazy_lru.remove(key)
                self._pg_lazy_lru.append(key)
            return key
        if note:
            seen = self._pg_lazy_seen.get(key, 0) + 1
            if seen >= 2:
                self._capture_lazy(key)
                return key
            if len(self._pg_lazy_seen) >= 256:  # a serving engine runs for days: forget the oldest sightings
                for k in list(self._pg_lazy_seen)[:128]:
                    del self._pg_lazy_seen[k]
            self._pg_lazy_seen[key] = seen
        return None

    def _stage_prefill_static(self, seqs: list[Sequence], tb: int, sb: int) -> int:
        """prefill_meta(seqs) at the static buffer's fixed offsets, padded to the bucket (pad tokens: id 0, position 0,
        slot -1; pad sequences: empty), the sampler's temperatures and {seed, step}; ONE async copy.  Returns the index
        of the pinned buffer (= of the token landing buffer and events) used."""
        self._pflip ^= 1
        h = self.pg_hosts[self._pflip]
        real_t = 0
        if seqs:
            m = batch_meta.prefill_meta(seqs, self.block_size, skip_cached=self.config.prefix_aware_prefill)
            real_t, n = int(m.cu_seqlens_q[-1]), len(seqs)
            h["ids"][:real_t], h["pos"][:real_t], h["slots"][:real_t] = m.input_ids, m.positions, m.slot_mapping
            h["cu_q"][:n + 1], h["cu_k"][:n + 1], h["kv_lens"][:n] = m.cu_seqlens_q, m.cu_seqlens_k, m.kv_lens
            h["cu_q"][n + 1:sb + 1], h["cu_k"][n + 1:sb + 1], h["kv_lens"][n:sb] = real_t, int(m.cu_seqlens_k[-1]), 0
            h["tables"][:sb] = -1
            h["tables"][:n, :m.block_tables.shape[1]] = m.block_tables
            h["temps"][:n] = [0.0 if s.greedy else s.temperature for s in seqs]
            h["temps"][n:sb] = 0.0
        else:
            # capture / its warm-up run: sb sequences sharing the tb tokens evenly that store nothing (slots -1) and
            # attend the reserved dummy block (the decode graphs' pad slot) - every kernel runs over mapped memory,
            # nothing is written.  (capture_prefill_graphs skips buckets whose share exceeds max_model_len.)
            edges = [tb * i // sb for i in range(sb + 1)]
            h["cu_q"][:sb + 1] = h["cu_k"][:sb + 1] = edges
            h["kv_lens"][:sb] = np.diff(edges)
            h["temps"][:sb] = 0.0
            h["tables"][:sb] = -1
            h["tables"][:sb, :-(-max(np.diff(edges)) // self.block_size)] = self.config.num_kvcache_blocks - 1
        h["ids"][real_t:tb], h["pos"][real_t:tb], h["slots"][real_t:tb] = 0, 0, -1
        rng = h["rng"].view(np.uint64)
        rng[0] = self.sampler.seed & 0xFFFFFFFFFFFFFFFF
        rng[1] = (self.sampler.step + 1) & 0xFFFFFFFFFFFFFFFF
        self._upload(self.pg_dev_buf, self.pg_host_bufs[self._pflip])
        return self._pflip

    def _bucket_for(self, n: int) -> int | None:
        for b in sorted(self.graphs):
            if b >= n:
                return b
        return None

    # ------------------------------------------------------------------ one step
    @torch.inference_mode()
    def run_model(self, input_ids: torch.Tensor, positions: torch.Tensor, is_prefill: bool, bucket: int | None):
        if is_prefill or bucket is None or bucket not in self.graphs:
            return self.model.compute_logits(self.model(input_ids, positions))
        self.graphs[bucket].replay()
        if self.world_size > 1 and bucket in self.graph_samples and not self.gather_logits:
            return self.graph_logits[bucket]  # this rank's vocabulary shard; the tokens are already picked
        return self.model.lm_head.gather(self.graph_logits[bucket])

    @torch.inference_mode()
    def run(self, seqs: list[Sequence], is_prefill: bool) -> list[int] | None:
        if not seqs:  # everything got preempted this step (reference would crash)
            return []
        real = len(seqs)
        if is_prefill and self.prefill_graphs:
            bucket = self._prefill_bucket(seqs, note=True)
            if bucket is not None:  # (the synchronous loop replays the captured steps too)
                return self.collect_prefill(self.launch_prefill(seqs, bucket))
        if is_prefill:
            input_ids, positions = self.prepare_prefill(seqs)
            bucket = None
            temps = self.prepare_sample(seqs) if self.rank == 0 else None
        else:
            bucket = self._bucket_for(real) if self.graphs else None
            input_ids, positions = self.prepare_decode(seqs, bucket)
            temps = self.dev["temps"][:real]
        logits = self.run_model(input_ids, positions, is_prefill, bucket)
        self.last_logits = logits  # debugging / parity hook (a reference, not a copy)
        tokens = None
        sampled_in_graph = not is_prefill and bucket in self.graphs and bucket in self.graph_samples
        if self.rank != 0:
            self.sampler.step += 1  # in lockstep with rank 0, which samples exactly once per step
        if self.rank == 0:
            if sampled_in_graph:
                self.sampler.step += 1  # the replayed graph sampled with this step (see _fill_decode_stage)
            else:
                self.sampler(logits, temps, out=self.tokens_dev[:real])
            landing = self.tokens_hosts[self._flip]
            landing[:real].copy_(self.tokens_dev[:real], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            tokens = landing[:real].tolist()
        self._steps_run += 1
        if self.xgmi is not None:  # every step: tokens computed behind a timed-out exchange must never be returned
            self._check_xgmi()
        reset_context()
        return tokens

    # ------------------------------------------------------------------ lookahead decode (one GPU)
    def can_launch_decode(self, n: int) -> bool:
        """A decode step of n sequences can be queued without waiting for its tokens: its graph ends in the
        token choice, so the step after it can read its input ids on the device."""
        return self._bucket_for(n) in self.graph_samples

    @property
    def max_launch_rows(self) -> int:
        return max(self.graph_samples, default=0)

    @torch.inference_mode()
    def launch_decode(self, seqs: list[Sequence], src_rows=None):
        """Queue one decode step and return a handle for collect().  src_rows[i] >= 0 names the row of the
        step queued just before whose sampled token is row i's input (not on the host yet)."""
        real = len(seqs)
        bucket = self._bucket_for(real)
        assert bucket in self.graph_samples
        self.lookahead_launches = getattr(self, "lookahead_launches", 0) + (src_rows is not None)
        if self.rank != 0 and self._events_recorded[self._flip ^ 1]:
            # a worker never waits for a step's tokens: before the pinned staging buffer of the step before last is
            # overwritten, make sure its upload has executed (rank 0 gets the same guarantee from collect())
            self.step_events[self._flip ^ 1].synchronize()
        b = self._fill_decode_stage(seqs, bucket, src_rows)
        self.graphs[bucket].replay()
        self.sampler.step += 1  # the graph sampled with this step (see _fill_decode_stage)
        if self.rank == 0:
            self._download_tokens(self.tokens_hosts[b], real)
            if self.xgmi is not None:  # the exchange's timeout flag travels with the tokens: no device sync here
                self.xgmi.status_async(self._xgmi_flag)
        self.step_events[b].record()
        self._events_recorded[b] = True
        self.last_logits = self.graph_logits[bucket]
        self._steps_run += 1
        return (b, real)

    # ------------------------------------------------------------------ queued prefill (one GPU)
    @property
    def can_launch_prefill(self) -> bool:
        """A prefill step can be queued without waiting for its tokens (launch_prefill / collect_prefill).  One GPU without
        a process group only: the TP workers' prefill ends in the exchange-status check of run(), and a step that issues
        RCCL collectives (also the one-rank group of MI355_TP1_COLLECTIVES) is waited for with the stream
        synchronisation of run()."""
        return self.world_size == 1 and not self.collective

    def prefill_launchable(self, seqs: list[Sequence]) -> bool:
        """May THIS prefill step be queued without waiting for its tokens?  One GPU: any step (eagerly if it has no
        graph).  Tensor parallelism: a step that replays a captured graph - rank 0 publishes `launch_prefill`, every
        rank stages the same metadata at its static addresses and replays the same graph, whose collectives are exchange
        kernels and whose last launch leaves the step's tokens on every rank (an eager TP step ends in RCCL calls and the
        logits gather: it stays synchronous, run())."""
        if self.can_launch_prefill:
            return True
        return self.world_size > 1 and bool(self.prefill_graphs) and self._prefill_bucket(seqs) is not None

    @torch.inference_mode()
    def launch_prefill(self, seqs: list[Sequence], bucket=False):
        """run(seqs, True) without its last step: metadata upload, the model, the sampler and the token copy are queued
        on the stream, an event marks their end; collect_prefill() waits for it.  The engine queues the NEXT prefill
        step between the two (Scheduler.lookahead_prefill) - `last_logits` (the parity hook) then names the logits of
        the step launched LAST, not of the step just collected."""
        real = len(seqs)
        if bucket is False:  # (a large step may be captured here, on its key's second sighting: before the clock starts)
            bucket = self._prefill_bucket(seqs, note=True)
        self.prefill_starts[self._pflip ^ 1].record()  # (prepare_prefill / _stage_prefill_static flip to this buffer)
        if bucket is not None:  # a captured step: stage at the fixed addresses, replay, the tokens are picked in the graph
            nb = self._pflip ^ 1
            if self.rank != 0 and self._prefill_events_recorded[nb]:
                # a worker never waits for a step's tokens: before the pinned mirror of the step before last is
                # overwritten, make sure its upload has executed (rank 0 gets that from collect_prefill)
                self.prefill_events[nb].synchronize()
            b = self._stage_prefill_static(seqs, bucket[0], bucket[1])
            self.prefill_graphs[bucket].replay()
            self.sampler.step += 1  # the graph sampled with this step (see _stage_prefill_static)
            self.prefill_graph_replays += 1
            self.last_logits = self.prefill_graph_logits[bucket][:real]
            if self.rank == 0:
                self._download_tokens(self.prefill_tokens_hosts[b], real)
                if self.xgmi is not None:  # the exchange's timeout flag travels with the tokens: no device sync here
                    self.xgmi.status_async(self._xgmi_flag)
            self.prefill_events[b].record()
            self._prefill_events_recorded[b] = True
            self._steps_run += 1
            return (b, real)
        assert self.world_size == 1, "an eager prefill step of tensor-parallel ranks is run(), not launch_prefill()"
        input_ids, positions = self.prepare_prefill(seqs)
        b = self._pflip
        temps = self.prepare_sample(seqs)
        logits = self.run_model(input_ids, positions, True, None)
        self.last_logits = logits
        self.sampler(logits, temps, out=self.tokens_dev[:real])
        self.prefill_tokens_hosts[b][:real].copy_(self.tokens_dev[:real], non_blocking=True)
        self.prefill_events[b].record()
        self._steps_run += 1
        reset_context()
        return (b, real)

    def prefill_done(self, handle) -> bool:
        """Have the queued step's tokens already reached the host?  (No wait.)"""
        return self.prefill_events[handle[0]].query()

    def collect_prefill(self, handle) -> list[int]:
        b, real = handle
        self.prefill_events[b].synchronize()
        if self._xgmi_flag is not None and int(self._xgmi_flag[0]):
            self._abort_workers()
            raise RuntimeError(f"rank 0: xGMI exchange timed out waiting for a peer; results are invalid - {self.xgmi.timeout_info()}")
        return self.prefill_tokens_hosts[b][:real].tolist()

    def prefill_device_ms(self, handle) -> float:
        """Device time of a collected prefill step, metadata upload to token copy (HIP events on the launch stream)."""
        b = handle[0]
        return self.prefill_starts[b].elapsed_time(self.prefill_events[b])

    def collect(self, handle) -> list[int]:
        b, real = handle
        self.step_events[b].synchronize()
        if self._xgmi_flag is not None and int(self._xgmi_flag[0]):
            self._abort_workers()
            raise RuntimeError(f"rank 0: xGMI exchange timed out waiting for a peer; results are invalid - {self.xgmi.timeout_info()}")
        return self.tokens_hosts[b][:real].tolist()

    def _check_xgmi(self):
        """The exchange kernel gives up on a peer after ~1 minute instead of hanging the GPU; what it
        returned then is not a sum.  Surface that as an error (checked after every step - the stream has just
        been synchronised for the token copy - and at exit)."""
        if self.xgmi is not None and self.xgmi.timed_out():
            self._abort_workers()
            raise RuntimeError(f"rank {self.rank}: xGMI all-reduce timed out waiting for a peer; results are invalid - "
                               f"{self.xgmi.timeout_info()}")

    def _abort_workers(self):
        """Rank 0 is about to raise: tell the workers to leave their receive loops (they kept spinning)."""
        if self.channel is not None and self.rank == 0 and not getattr(self, "_aborted", False):
            self._aborted = True
            self.channel.send("abort")
