"""World size 2 on the ONE leased GPU (both ranks on cuda:0, `gloo` process group): the N > 1 code of BASELINE config 3
and of both bench scripts with real second ranks -- launched exactly as the driver launches the N-GPU runs
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`).  RCCL itself needs one
device per rank, so the wire is gloo here; the RCCL wire runs at world size 1 in tests/test_gpu_configs.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun2(script_args, port, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port)] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]        # rank 0 prints, rank 1 stays silent
    return json.loads(lines[0])


def test_ddp_two_ranks_gradients():
    """tests/ddp_world2_worker.py: DDP over the HIP path with two ranks; identical gradients on both ranks, equal to
    the mean of the ranks' own gradients and to the whole-batch gradient within 16-bit accumulation noise."""
    line = _torchrun2(['tests/ddp_world2_worker.py'], 29541)
    print(line)
    assert line['world'] == 2 and line['params'] > 20


def test_bench_two_ranks():
    """bench.py --gpus 2: weak scaling, value = the tokens of BOTH ranks over the slower rank's time, one JSON line,
    no cpu_baseline leg at N > 1; the auto batch is rank 0's pick, broadcast."""
    line = _torchrun2(['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--dist-backend',
                       'gloo'], 29542)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak'
    assert line['config']['batch_per_gpu'] == 4 and line['config']['global_batch'] == 8
    tokens = 2 * 4 * 1024 * 2
    assert abs(line['value'] - tokens / (line['ms_per_step'] * 2e-3)) <= 1e-3 * line['value']
    assert 'cpu_baseline' not in line
    line = _torchrun2(['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', 'auto',
                       '--batch-candidates', '2,4', '--dist-backend', 'gloo'], 29543)
    assert line['n_gpus'] == 2 and line['config']['batch_per_gpu'] in (2, 4)


def test_train_step_two_ranks():
    """scripts/bench_train_step.py with two DDP ranks: Backpack-Small at seq 1024, the reference recipe (AMP over fp32
    parameters, dropout 0.1, fused AdamW), 682 MB of fp32 gradients all-reduced per step."""
    line = _torchrun2(['scripts/bench_train_step.py', '--batch', '1', '--steps', '1', '--warmup', '1', '--dist-backend',
                       'gloo'], 29544)
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['launch'].endswith('gloo')
    assert line['grad_allreduce_bytes'] > 680e6 and 5 < line['loss'] < 13
    # what an N-GPU run needs to be conclusive: the collective timed on its own, as bus bandwidth, next to both bounds
    assert line['allreduce_ms'] > 0 and line['bus_gbps'] > 0 and line['ms_per_step_no_sync'] > 0
    assert abs(line['bound_ring_ms'] - 2 * 0.5 * line['grad_allreduce_bytes'] / 153e9 * 1e3) < 1e-2
    assert abs(line['bound_all_links_ms'] - line['bound_ring_ms']) < 1e-6      # N = 2: one peer, one link
    # the reference's gradient-compression hook (fp16 on the wire: half the bytes), same loss
    half = _torchrun2(['scripts/bench_train_step.py', '--batch', '1', '--steps', '1', '--warmup', '1', '--dist-backend',
                       'gloo', '--grad-compress', 'fp16'], 29545)
    assert half['grad_compress'] == 'fp16' and half['grad_allreduce_bytes'] * 2 == line['grad_allreduce_bytes']
    assert abs(half['loss'] - line['loss']) < 0.05
