"""Headline benchmark: forward tokens/s of Backpack-Small (d=768, 12 heads, 12 layers, k=16 senses)
at seq 1024, bf16, on N MI355X (BASELINE.json `metric`), plus the roofline fraction of the
dominant HIP kernel and the CPU-eager baseline timed in the same run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B|auto] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full forward (ids -> logits) over one batch of synthetic token ids already resident
in HBM; weights are random with the reference's init scheme (no network for checkpoints).
The forward shards by batch with no data-path collective: N ranks = N independent replicas
(weak scaling), value = all ranks' tokens / max-over-ranks time.

Per-kernel durations are measured live with HIP events (torch.cuda.Event on the stream the kernels
are launched on = torch's current stream) around every launch inside the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'backpacks-flash-attn_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# the host driver of this pool only supports dmabuf IPC: without it RCCL's cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument` at world size > 1 (the image exports it; a bare launcher may not)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402

# MI355X peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16 MFMA
PEAK_HBM_GBPS = 8000.0      # HBM3E
MAX_SCLK_MHZ = 2400.0       # the clock the MFMA peak is quoted at

WORKLOADS = {
    # name: (model, seq, dtype, default batch)  -- BASELINE.json configs[1] is the headline
    'small-1024': ('small', 1024, 'bf16', 64),
    'small-4096-fp16': ('small', 4096, 'fp16', 8),
    'mini-k64-1024': ('mini-k64', 1024, 'bf16', 32),
    # the few-sense ends of the reference's ablation (training/configs/experiment/owt/backpack-mini-flash-vecs-4.yaml, -vecs-1):
    # wide senses, csrc/sense_wide.hip
    'mini-k4-1024': ('mini-k4', 1024, 'bf16', 32),
    'mini-k1-1024': ('mini-k1', 1024, 'bf16', 32),
    'micro-128': ('micro', 128, 'bf16', 4),
}
MODELS = {
    'micro': dict(n_embd=384, n_head=6, n_layer=6, num_content_vectors=16),
    'mini-k64': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=64, shrink_final_inner=True),
    'mini-k4': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=4, shrink_final_inner=True),
    'mini-k1': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=1, shrink_final_inner=True),
    'small': dict(n_embd=768, n_head=12, n_layer=12, num_content_vectors=16),
}


class KernelClock:
    """HIP-event stopwatch around individual kernel launches."""

    def __init__(self):
        self.enabled = False
        self.spans = {}

    def wrap(self, name, fn):
        def timed(*a, **kw):
            if not self.enabled:
                return fn(*a, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.spans.setdefault(name, []).append((e0, e1))
            return out
        return timed

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, spans in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in spans]
            out[name] = dict(launches=len(ms), avg_ms=sum(ms) / len(ms), total_ms=sum(ms))
        return out


class ClockPowerSampler:
    """Shader clock and socket power of one GPU, sampled on a side thread while a region runs (review round 5: "power
    limited" must be checkable from the driver-run line).  Source, in order: the `amdsmi` Python binding that ships with
    ROCm (gpu_metrics: per-XCD `current_gfxclks`, `current_socket_power`), `amdsmi_get_clock_info` / `..._power_info`,
    sysfs (`pp_dpm_sclk` + hwmon `power1_average`).  Nothing readable -> `source: None` and null means (never a guess)."""

    def __init__(self, device_index=0, period_s=0.02):
        import threading
        self.period = period_s
        self.index = device_index
        self._stop = threading.Event()
        self._thread = None
        self.samples = []
        self.source = None
        self.power_cap_w = None
        self._read = self._pick_reader()

    # -- readers: each returns (sclk_mhz or None, power_w or None) -----------------------------------------------------
    def _pick_reader(self):
        try:
            import amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception:
                pass
            handles = amdsmi.amdsmi_get_processor_handles()
            h = handles[self.index if self.index < len(handles) else 0]
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h)
                c = cap.get('power_cap')
                if isinstance(c, (int, float)) and c > 0:
                    self.power_cap_w = c / 1e6 if c > 1e5 else float(c)
            except Exception:
                pass

            def num(x):
                return float(x) if isinstance(x, (int, float)) and 0 < x < 65535 else None

            def metrics():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clks = [num(c) for c in (m.get('current_gfxclks') or [])]
                clks = [c for c in clks if c]
                sclk = sum(clks) / len(clks) if clks else num(m.get('current_gfxclk'))
                pw = num(m.get('current_socket_power')) or num(m.get('average_socket_power'))
                return sclk, pw

            def infos():
                ct = getattr(amdsmi.AmdSmiClkType, 'GFX', amdsmi.AmdSmiClkType.SYS)
                c = amdsmi.amdsmi_get_clock_info(h, ct)
                pw = amdsmi.amdsmi_get_power_info(h)
                return num(c.get('clk')), (num(pw.get('current_socket_power')) or num(pw.get('average_socket_power'))
                                           or num(pw.get('socket_power')))

            for name, fn in (('amdsmi gpu_metrics (mean of current_gfxclks over the XCDs, current_socket_power)', metrics),
                             ('amdsmi_get_clock_info(GFX).clk, amdsmi_get_power_info', infos)):
                try:
                    a, b = fn()
                    if a or b:
                        self.source = name
                        return fn
                except Exception:
                    continue
        except Exception:
            pass
        import glob
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/pp_dpm_sclk'))
        if cards:
            dev = os.path.dirname(cards[self.index if self.index < len(cards) else 0])
            hw = sorted(glob.glob(os.path.join(dev, 'hwmon', 'hwmon*', 'power1_average'))
                        + glob.glob(os.path.join(dev, 'hwmon', 'hwmon*', 'power1_input')))

            def sysfs():
                sclk = pw = None
                for ln in open(os.path.join(dev, 'pp_dpm_sclk')):
                    if '*' in ln:
                        sclk = float(ln.split(':')[1].lower().replace('mhz', '').replace('*', '').strip())
                if hw:
                    pw = float(open(hw[0]).read()) / 1e6
                return sclk, pw
            try:
                a, b = sysfs()
                if a or b:
                    self.source = 'sysfs pp_dpm_sclk (current level) + hwmon power1_average'
                    return sysfs
            except Exception:
                pass
        return None

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        import threading
        self.samples = []
        self._stop.clear()
        if self._read is not None:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        clk = [a for a, _ in self.samples if a]
        pw = [b for _, b in self.samples if b]
        return dict(sclk_mhz_mean=round(sum(clk) / len(clk), 1) if clk else None,
                    sclk_mhz_min=round(min(clk), 1) if clk else None, sclk_mhz_max=round(max(clk), 1) if clk else None,
                    power_w_mean=round(sum(pw) / len(pw), 1) if pw else None,
                    power_w_max=round(max(pw), 1) if pw else None, power_cap_w=self.power_cap_w,
                    samples=len(self.samples), period_ms=round(self.period * 1e3, 1), source=self.source)


def instrument(clock):
    """Bracket each HIP kernel launch with events.  The fused sense mix is split into its two
    launches (LSE pre-pass, mix) through the public lse= argument so each kernel is timed alone."""
    import bp_hip
    raw_flash, raw_lse, raw_mix = bp_hip.flash_fwd, bp_hip.sense_lse, bp_hip.sense_mix
    t_flash = clock.wrap('flash_fwd_kernel', raw_flash)
    t_lse = clock.wrap('flash_fwd_kernel[lse-only,senses]', raw_lse)
    t_mix = clock.wrap('sense_mix_kernel', raw_mix)

    def mix_two_launches(qk, content, softmax_scale=None, out=None, lse=None, key_weight=None):
        if lse is None:
            lse = t_lse(qk, softmax_scale)
        return t_mix(qk, content, softmax_scale, out=out, lse=lse, key_weight=key_weight)

    raw_gather = bp_hip.sense_mix_gather
    t_gather = clock.wrap('sense_mix_kernel', raw_gather)

    def gather_two_launches(qk, table, row_index, softmax_scale=None, out=None, lse=None):
        if lse is None:
            lse = t_lse(qk, softmax_scale)
        return t_gather(qk, table, row_index, softmax_scale, out=out, lse=lse)

    bp_hip.flash_fwd = t_flash
    bp_hip.sense_mix = mix_two_launches
    bp_hip.sense_mix_gather = gather_two_launches
    bp_hip.add_layer_norm = clock.wrap('add_layer_norm_kernel', bp_hip.add_layer_norm)


def sustained_flash_probe(cfg, batch, seq, dtype, device, device_index, min_seconds=1.0):
    """The trunk attention launch alone, back to back for >= `min_seconds` at the bench batch, with the clock / power
    sampler running: the shader clock the chip SUSTAINS under this kernel (the step-level mean mixes it with the GEMMs).
    Random q, k, v of the model's shapes (zeros would clock higher: MI355X_MICROARCH, DVFS give-back)."""
    import bp_hip
    h, dh = cfg.n_head, cfg.n_embd // cfg.n_head
    try:
        qkv = torch.randn(batch * seq, 3, h, dh, device=device, dtype=dtype)
        o = torch.empty(batch * seq, h, dh, device=device, dtype=dtype)
    except torch.OutOfMemoryError:
        return None

    def launch():
        bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], o, None, None, seq, seq, dh ** -0.5, True)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    n = max(8, int(min_seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    smi = ClockPowerSampler(device_index, period_s=0.01).start()
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    res = smi.stop()
    res.update(launches=n, avg_launch_ms=round(e0.elapsed_time(e1) / n, 4))
    return res


def build_model(name, seq, dtype, device):
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    cfg = BackpackConfig(vocab_size=50257, n_positions=seq, scale_attn_by_inverse_layer_idx=True,
                         use_flash_attn=True, fused_bias_fc=True, fused_dense_gelu_dense=True,
                         fused_dropout_add_ln=True, pad_vocab_size_multiple=8, **MODELS[name])
    torch.manual_seed(0)
    model = BackpackLMHeadModel(cfg, device=device, dtype=dtype).eval()
    return cfg, model


def algorithmic_work(cfg, batch, seq):
    """SURVEY.md section 8(d) per-sample figures x batch.  2-byte elements, causal pairs only."""
    d, h, L, k = cfg.n_embd, cfg.n_head, cfg.n_layer, cfg.num_content_vectors
    pairs = seq * (seq + 1) // 2
    return {
        # one trunk layer = one launch: 4*pairs*d flops, q,k,v read + o written = 8*S*d bytes
        'flash_fwd_kernel': dict(flops=4 * pairs * d * batch, bytes=8 * seq * d * batch),
        # fused mix launch: QK^T once + alpha.C ; reads qk (4*S*d) + C (2*k*S*d), writes out (2*S*d)
        'sense_mix_kernel': dict(flops=2 * pairs * d * (1 + k) * batch,
                                 bytes=(4 + 2 * k + 2) * seq * d * batch),
        # LSE pre-pass: QK^T once, reads qk, writes k*S fp32
        'flash_fwd_kernel[lse-only,senses]': dict(flops=2 * pairs * d * batch,
                                                  bytes=(4 * seq * d + 4 * k * seq) * batch),
        # fused add + LayerNorm: read x0 (2 B) + residual (4 B), write residual (4 B) + z (2 B) per element;
        # ~8 flops per element.  HBM-bound.
        'add_layer_norm_kernel': dict(flops=8 * seq * d * batch, bytes=12 * seq * d * batch),
    }


class LogitsBuffer:
    """ONE persistent (max batch, S, vocab) block the LM head writes into (`logits_out=`): at the HBM-filling batch the
    logits are most of the memory (Small, S = 1024: 103 MB per sample), and a fresh 100-200 GiB allocation per step is
    what the caching allocator failed to re-place in round 3 (r03_n / r03_s).  Smaller batches use a leading slice."""

    def __init__(self, vocab, seq, dtype, device):
        self.vocab, self.seq, self.dtype, self.device = vocab, seq, dtype, device
        self.buf = None

    def bytes_per_sample(self):
        return self.seq * self.vocab * torch.empty((), dtype=self.dtype).element_size()

    def get(self, batch):
        if self.buf is None or self.buf.shape[0] < batch:
            self.buf = None
            torch.cuda.empty_cache()
            self.buf = torch.empty((batch, self.seq, self.vocab), dtype=self.dtype, device=self.device)
        return self.buf[:batch]


def pick_batch(model, make_ids, candidates, seq, device, logits, steps=3, hbm_limit=0.90):
    """Untimed-region batch sweep ("B swept ... to the max that fits"): tokens/s of `steps`
    forwards at each candidate batch after 2 warm-up forwards, logits written into the persistent block.  The first
    (smallest) candidate measures the footprint per sample besides the logits; the block is then allocated ONCE for the
    largest candidate whose estimated peak stays under `hbm_limit` of HBM, and every candidate up to it runs.  The
    LARGEST batch within 1 % of the best rate wins (the curve is flat once the GEMMs are at their rate, so this is the
    HBM-filling batch).  Returns (batch, the sweep table)."""
    table = []
    total_mem = torch.cuda.get_device_properties(device).total_memory
    candidates = sorted(candidates)
    per_sample_other, fixed = None, torch.cuda.memory_allocated(device)
    largest = candidates[0]
    for b in candidates:
        if per_sample_other is None:
            largest = b
        if per_sample_other is not None and b > largest:
            est = fixed + (per_sample_other + logits.bytes_per_sample()) * b
            table.append(dict(batch=b, ms_per_step=None, tokens_per_s=0.0, peak_mem_gb=None,
                              note=f'skipped: ~{est / 2**30:.0f} GiB estimated'))
            continue
        try:
            out = logits.get(b if per_sample_other is None else largest)[:b]
            torch.cuda.reset_peak_memory_stats(device)
            ids = make_ids(b)
            with torch.no_grad():
                for _ in range(2):
                    model(ids, logits_out=out)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    model(ids, logits_out=out)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            peak = torch.cuda.max_memory_allocated(device)
            # (the footprint is taken from the first candidate large enough for the deduplicated content network --
            # smaller ones also hold the per-position content tensor, 25 MB per sample at Small)
            vocab = model.lm_head.weight.shape[0]
            if per_sample_other is None and (b * seq >= vocab or b == candidates[-1]):
                per_sample_other = max(peak - fixed - logits.bytes_per_sample() * b, 0) / b
                fit = [c for c in candidates
                       if fixed + (per_sample_other + logits.bytes_per_sample()) * c <= hbm_limit * total_mem]
                largest = max(fit) if fit else b
            # what THIS batch needs: the peak minus the part of the persistent logits block (sized for the largest
            # candidate) that this batch does not write
            block = logits.buf.numel() * logits.buf.element_size() if logits.buf is not None else 0
            used = peak - max(block - logits.bytes_per_sample() * b, 0)
            table.append(dict(batch=b, ms_per_step=round(dt * 1e3, 3), tokens_per_s=round(b * seq / dt, 1),
                              peak_mem_gb=round(used / 2**30, 1), hbm_frac=round(used / total_mem, 3),
                              allocated_gb=round(peak / 2**30, 1)))
        except torch.OutOfMemoryError:
            table.append(dict(batch=b, ms_per_step=None, tokens_per_s=0.0, peak_mem_gb=None, note='out of HBM'))
        finally:
            ids = out = None
    ran = [r for r in table if r['tokens_per_s']]
    if not ran:
        raise SystemExit('no candidate batch fits in HBM')
    best = max(r['tokens_per_s'] for r in ran)
    pick = max(r['batch'] for r in ran if r['tokens_per_s'] >= 0.99 * best)
    return pick, table


def cpu_baseline(model_name, seq, budget_s=15.0, threads=None):
    """The reference's eager CPU path (restated in oracle/ref_cpu.py, validated against the real
    reference by tests/golden/make_golden.py) on this host's cores, fp32.
    Threads: the cores this process may run on, capped at 16 -- measured on the MI355X host
    (256 logical CPUs, scripts/cpu_threads_probe.py): 8 thr 0.90 s, 16 thr 0.72 s, 32 thr 0.87 s,
    64 thr 1.53 s, 128 thr 4.2 s, 256 thr 36 s per Small forward: torch's eager ops get slower past a
    couple of dozen threads on these small-per-op workloads."""
    from oracle import ref_cpu as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = threads or min(avail, 16)
    torch.set_num_threads(cores)
    cfg = R.make_config(model_name, n_positions=seq)
    sd = R.init_state_dict(cfg, seed=0)
    b = 1
    ids = torch.randint(0, 50257, (b, seq), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        R.backpack_forward(sd, cfg, ids)  # warm-up
        times = []
        t_start = time.perf_counter()
        while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 50):
            t0 = time.perf_counter()
            R.backpack_forward(sd, cfg, ids)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 2 * budget_s:
                break
    mean = sum(times) / len(times)
    # BASELINE.json configs[0] exactly (the reference's own CPU-runnable case): Micro, B=4, S=128, fp32
    mcfg = R.make_config('micro', n_positions=128)
    msd = R.init_state_dict(mcfg, seed=0)
    mids = torch.randint(0, 50257, (4, 128), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        R.backpack_forward(msd, mcfg, mids)
        t0 = time.perf_counter()
        for _ in range(5):
            R.backpack_forward(msd, mcfg, mids)
        micro = (time.perf_counter() - t0) / 5
    return dict(value=b * seq / mean, unit='tokens/s', cores=cores, kind='port',
                sample=f'oracle/ref_cpu.backpack_forward (reference eager path), Backpack-{model_name} '
                       f'fp32, batch {b} x seq {seq}, {len(times)} timed forwards after 1 warm-up, '
                       f'mean {mean:.3f} s/forward, torch {torch.get_num_threads()} threads',
                config1_micro_b4_s128_tokens_per_s=round(4 * 128 / micro, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', default='auto',
                    help="samples per GPU per step, or 'auto' (default): time a few steps at each of "
                         "--batch-candidates and keep the fastest (north_star: batch sized for 288 GB of HBM)")
    ap.add_argument('--batch-candidates', default=None,
                    help='comma-separated batch sizes tried by --batch auto (default: per workload)')
    ap.add_argument('--workload', default='small-1024', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=None)
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--eager-senses', action='store_true',
                    help='comparison runs only: sense weights and combination of a WIDE-sense model (d_k > 128) as the '
                         "reference's eager op sequence on the GPU instead of csrc/sense_wide.hip (what rounds 1-5 ran)")
    ap.add_argument('--no-clock-probe', action='store_true',
                    help='skip the sustained run of the dominant kernel alone that measures its shader clock / power')
    ap.add_argument('--content', default=None, choices=['batch', 'cached', 'position'],
                    help="how the timed step gets its sense vectors (src/models/backpack.py, BackpackModel.sense_table_mode): "
                         "'batch' (default for eager launches) = content network once per distinct token id of the batch, the "
                         "table rebuilt EVERY step; 'cached' (default with --graph: the only per-token form a graph can hold) = "
                         "the whole-vocabulary table built once per weight version, outside the timed region; 'position' = "
                         "the reference's order, every position through the content network.  The other two orders are timed "
                         "for a few steps in the same process and reported next to `value`.")
    ap.add_argument('--no-content-dedup', action='store_true', help="same as --content position")
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'nccl' (= RCCL, one rank per GPU) for every real run; 'gloo' moves the same collectives "
                         "through host memory and lets several ranks share one GPU -- the test-suite's world-size-2 run "
                         "on the one leased GPU")
    ap.add_argument('--graph', action='store_true',
                    help='capture one forward in a HIP graph and replay it per step (launch-bound small '
                         'configs; per-kernel events are not taken in this mode)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    if args.dist_backend == 'gloo':      # test-suite only: ranks may share a device
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    # under torch.distributed.run (any world size, 1 included: the launch path is then the one the N-GPU runs take)
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):
        import torch.distributed as dist
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)   # 'nccl' is RCCL on ROCm
        else:
            dist.init_process_group('gloo')

    import bp_hip
    bp_hip.lib()   # fail loudly if the HIP extension is missing -- no fallback path exists
    if args.eager_senses:
        bp_hip.SENSE_MAX_DK = 128

    model_name, seq, dtype_name, default_batch = WORKLOADS[args.workload]
    dtype = torch.bfloat16 if dtype_name == 'bf16' else torch.float16
    cfg, model = build_model(model_name, seq, dtype, device)
    MODES = {'batch': 'batch', 'cached': 'cached', 'position': 'off'}
    content = 'position' if args.no_content_dedup else (args.content or ('cached' if args.graph else 'batch'))
    if args.graph and content == 'batch':
        raise SystemExit("--graph cannot hold --content batch (torch.unique has a data-dependent shape)")
    model.transformer.sense_table_mode = MODES[content]

    def make_ids(b):
        return torch.randint(0, 50257, (b, seq), device=device,
                             generator=torch.Generator(device=device).manual_seed(1234 + rank))

    sweep = None
    logits = LogitsBuffer(cfg.vocab_size, seq, dtype, device)
    cands = []
    if args.batch == 'auto':
        cands = ([int(c) for c in args.batch_candidates.split(',')] if args.batch_candidates
                 else [default_batch * m for m in (1, 2, 4, 8, 16, 24, 26, 28, 30, 32)])
        batch, sweep = pick_batch(model, make_ids, cands, seq, device, logits)
        torch.cuda.synchronize()
        if dist is not None:    # every rank runs the SMALLEST batch any rank picked (identical GPUs pick alike)
            t = torch.tensor([batch], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            batch = int(t.item())
    else:
        batch = int(args.batch)
    ids = make_ids(batch)
    logits_out = logits.get(batch)[:batch]

    clock = KernelClock()
    if not args.no_kernel_events and not args.graph:
        instrument(clock)

    def eager_step():
        with torch.no_grad():
            return model(ids, logits_out=logits_out).logits

    step = eager_step
    if args.graph:
        # HIP graph: the whole forward (torch ops and the C-ABI launches alike go to the capture stream)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = eager_step()

        def step():
            graph.replay()
            return graph_out

    # Warm-up.  The logits live in the persistent block, so the step allocates only transient activations; should a
    # rank still run out of HBM, ALL ranks step down together to the next smaller candidate (a MIN all-reduce of a flag
    # per attempt: ranks must never time different batches).
    out = None
    while True:
        failed = 0
        try:
            for _ in range(args.warmup):
                out = step()
        except torch.OutOfMemoryError:
            failed = 1
        if dist is not None:
            t = torch.tensor([failed], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            failed = int(t.item())
        if not failed:
            break
        out = None
        smaller = [c for c in cands if c < batch]
        if args.graph or not smaller:
            raise SystemExit(f'out of HBM at batch {batch} and no smaller candidate to fall back to')
        batch = max(smaller)
        if sweep is not None:
            sweep.append(dict(batch=batch, note='timed run fell back to this batch: out of HBM at the picked one'))
        ids = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        ids = make_ids(batch)
        logits_out = logits.get(batch)[:batch]
    out = None
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    clock.enabled = True
    smi = ClockPowerSampler(local_rank).start() if rank == 0 else None   # side thread; reads the SMU's metrics table only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clock.enabled = False
    step_clocks = smi.stop() if smi is not None else None
    assert out.shape == (batch, seq, cfg.vocab_size) and bool(torch.isfinite(out[0, -1].float()).all())

    # The same step in the OTHER content orders (the reference's per-position order among them), timed next to the headline
    # so that all numbers come from one process on one box: a few steps, same batch, same barriers.
    hbm_peak = torch.cuda.max_memory_allocated(device)
    with torch.no_grad():
        # which order the timed steps actually ran ('batch' falls back to per position below vocab positions)
        if content == 'batch' and not model.transformer._dedup_applies(ids):
            content = 'position'
        if content == 'cached' and model.transformer.sense_table() is None:
            content = 'position'

    def time_other(mode):
        """tokens/s of `--steps` eager steps with sense_table_mode = `mode` at the headline batch, same barriers (the
        per-position order runs its content network and mix over chunks of samples, src/models/backpack.py, so its 25 MB
        per sample of content tensor no longer limit the batch; should a rank still run out of HBM: 3/4 of the batch,
        repeatedly, and the line says so in `batch_per_gpu`)."""
        model.transformer.sense_table_mode = MODES[mode]
        n = max(1, args.steps)
        b_o, result = batch, None
        while result is None:
            failed, el = 0, 0.0
            try:
                with torch.no_grad():
                    if mode == 'batch' and not model.transformer._dedup_applies(ids[:b_o]):
                        return dict(value=None, note='fewer positions than vocabulary entries: this order does not apply')
                    model(ids[:b_o], logits_out=logits_out[:b_o])
                torch.cuda.synchronize()   # (no collective inside the try: a rank that runs out of HBM must not leave
                t1 = time.perf_counter()   #  the others waiting in a barrier it never reaches)
                with torch.no_grad():
                    for _ in range(n):
                        model(ids[:b_o], logits_out=logits_out[:b_o])
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
            except torch.OutOfMemoryError:
                failed = 1
            if dist is not None:
                t = torch.tensor([failed], device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                failed = int(t.item())
            if failed:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                b_o = b_o * 3 // 4
                if b_o < 1:
                    return dict(value=None, note='out of HBM')
                continue
            if dist is not None:
                t = torch.tensor([el], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            result = dict(value=round(world * b_o * seq * n / el, 1), unit='tokens/s', steps=n,
                          ms_per_step=round(el / n * 1e3, 3), batch_per_gpu=b_o)
        return result

    others = {}
    out = None
    for mode in ('position', 'cached', 'batch'):
        if mode != content and not args.graph:   # (a replayed graph against eager launches would compare launch overheads)
            others[mode] = time_other(mode)
    model.transformer.sense_table_mode = MODES[content]

    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernels = clock.summary()
    work = algorithmic_work(cfg, batch, seq)
    kernel_rows = []
    for name, k in kernels.items():
        w = work[name]
        sec = k['avg_ms'] * 1e-3
        tf = w['flops'] / sec / 1e12
        gbs = w['bytes'] / sec / 1e9
        kernel_rows.append(dict(kernel=name, launches_per_step=k['launches'] // max(args.steps, 1),
                                avg_ms=round(k['avg_ms'], 4), total_ms=round(k['total_ms'], 3),
                                algorithmic_flops=w['flops'], algorithmic_bytes=w['bytes'],
                                tflops=round(tf, 1), mfma_frac=round(tf / PEAK_MFMA_TFLOPS, 4),
                                gbps=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBPS, 4)))
    kernel_rows.sort(key=lambda r: -r['total_ms'])

    if rank == 0:
        tokens = world * batch * seq * args.steps
        line = {
            'metric': 'tokens/sec fwd, Backpack-Small seq=1024' if args.workload == 'small-1024'
                      else f'tokens/sec fwd, {args.workload}',
            'value': round(tokens / elapsed, 1),
            'unit': 'tokens/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': dtype_name,
            'data': 'synthetic token ids, random weights (reference init scheme)',
            'launch': 'hip-graph replay' if args.graph else 'eager launches',
            **({'sense_path': 'eager op sequence (comparison run, --eager-senses)'} if args.eager_senses else {}),
            'config': {'workload': f'Backpack-{model_name} forward (ids -> logits), d={cfg.n_embd}, '
                                   f'{cfg.n_head} heads, {cfg.n_layer} layers, k={cfg.num_content_vectors} '
                                   f'senses, vocab {cfg.vocab_size}, seq {seq}',
                       'batch_per_gpu': batch, 'global_batch': batch * world, 'seq_len': seq,
                       'batch_choice': 'auto: largest batch within 1 % of the best rate in batch_sweep (candidates up to 90 % of HBM; logits in one persistent block)' if sweep else 'given',
                       'hbm_frac_peak': round(hbm_peak / torch.cuda.get_device_properties(device).total_memory, 3),
                       'content_network': {
                           'batch': 'once per distinct token id of the batch, EVERY step (torch.unique; the mix kernel gathers the '
                                    'rows; the sense vectors depend on the token alone)',
                           'cached': 'whole-vocabulary sense table built once per weight version OUTSIDE the timed region; '
                                     'the mix kernel gathers its rows by token id',
                           'position': 'once per position (the reference\'s order of operations)'}[content],
                       'parallelism': f'{world} independent batch replicas (no data-path collective)'},
        }
        # the other content orders, same process / batch: the reference's order of operations (content_per_position), the
        # cached whole-vocabulary table (content_cached_table), the per-step table of the batch's distinct ids
        for mode, key in (('position', 'content_per_position'), ('cached', 'content_cached_table'),
                          ('batch', 'content_per_batch_table')):
            if mode in others:
                line[key] = others[mode]
        if kernel_rows:
            # `roofline` = the attention-path kernel with the largest total time (the path BASELINE.json's
            # north_star names: flash attention tile / sense contraction); every timed kernel, including the
            # HBM-bound fused add+LayerNorm, is listed in `kernels`.
            hot = [r for r in kernel_rows if r['kernel'] != 'add_layer_norm_kernel'] or kernel_rows
            dom = hot[0]
            # HBM bytes per launch are NOT measured in this run (PMC counters need their own rocprofv3 pass):
            # the figure is quoted from profiles/traffic.json, which names the tracked PMC summary it was
            # computed from; null when no pass exists for this workload / batch.
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    key = f"{args.workload}/b{batch}/{dom['kernel']}"
                    traffic = tj.get(key)
                    if traffic is not None:
                        traffic_source = ('quoted from profiles/traffic.json (' + tj.get('_files', {}).get(key, '?') + '): '
                                          + tj.get('_source', 'rocprofv3 --pmc pass'))
                    else:
                        # no counter pass at exactly this batch (the auto pick moves between the candidates on the
                        # plateau): scale the nearest measured batch linearly -- per-sample traffic of these kernels is
                        # flat from 512 samples up (b512 / b1536 / b1920 / b2048 / b2304 entries of the table)
                        measured = sorted((abs(int(k.split('/')[1][1:]) - batch), int(k.split('/')[1][1:]), v)
                                          for k, v in tj.items() if k.startswith(args.workload + '/b')
                                          and k.endswith('/' + dom['kernel']) and isinstance(v, int))
                        if measured and batch >= 512 and measured[0][1] >= 512:
                            _, b_ref, v_ref = measured[0]
                            ref_key = f"{args.workload}/b{b_ref}/{dom['kernel']}"
                            traffic = int(v_ref * batch / b_ref)
                            traffic_source = (f'scaled x{batch}/{b_ref} from the batch-{b_ref} counter pass in profiles/traffic.json ('
                                              + tj.get('_files', {}).get(ref_key, '?') + '): '
                                              + tj.get('_source', 'rocprofv3 --pmc pass'))
                except Exception:
                    traffic = None
            if dom['kernel'] == 'add_layer_norm_kernel':
                line['roofline'] = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': dom['gbps'],
                                    'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': dom['hbm_frac'],
                                    'traffic': traffic, 'traffic_source': traffic_source,
                                    'avg_launch_ms': dom['avg_ms']}
            else:
                line['roofline'] = {'kernel': dom['kernel'], 'bound': 'mfma',
                                    'achieved': dom['tflops'], 'peak': PEAK_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': dom['mfma_frac'], 'traffic': traffic,
                                    'traffic_source': traffic_source, 'avg_launch_ms': dom['avg_ms'], 'hbm_gbps': dom['gbps'],
                                    'hbm_frac': dom['hbm_frac']}
            # the HBM-bound fused add + LayerNorm is excluded from `roofline` BY RULE (the path north_star names is the
            # attention tile / sense contraction) even when its total time is the largest of this repository's kernels:
            # its own roofline rides along here
            ln = next((r for r in kernel_rows if r['kernel'] == 'add_layer_norm_kernel'), None)
            if ln is not None and dom['kernel'] != 'add_layer_norm_kernel':
                line['roofline']['excluded_by_rule'] = {
                    'kernel': ln['kernel'], 'why': 'memory-bound glue around the attention path, not the path itself',
                    'bound': 'hbm', 'achieved': ln['gbps'], 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': ln['hbm_frac'],
                    'launches_per_step': ln['launches_per_step'], 'total_ms': ln['total_ms'], 'avg_launch_ms': ln['avg_ms']}
            # clock / power: (a) over the timed region (all kernels of the step), (b) under the dominant attention
            # launch alone, sustained.  frac_at_sustained_clock = achieved / (peak x sclk / 2400 MHz): what the launch
            # reaches of the matrix peak at the clock the chip actually gives it.
            if step_clocks is not None:
                line['step_clock_power'] = step_clocks
            if dom['kernel'] == 'flash_fwd_kernel' and not args.no_clock_probe:
                sus = sustained_flash_probe(cfg, batch, seq, dtype, device, local_rank)
                if sus is not None:
                    r = line['roofline']
                    r['sclk_mhz_mean'] = sus['sclk_mhz_mean']
                    r['power_w_mean'] = sus['power_w_mean']
                    r['power_cap_w'] = sus['power_cap_w']
                    if sus['sclk_mhz_mean']:
                        r['frac_at_sustained_clock'] = round(r['achieved'] / (PEAK_MFMA_TFLOPS * sus['sclk_mhz_mean'] / MAX_SCLK_MHZ), 4)
                    r['sustained_probe'] = dict(sus, what='this kernel alone, back to back at the bench batch on random q, k, v; '
                                                'sampler on a side thread')
            line['kernels'] = kernel_rows
        if sweep:
            line['batch_sweep'] = sweep
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model_name, seq, threads=args.cpu_threads)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
