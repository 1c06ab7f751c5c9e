"""Race / determinism stress for the ring kernels: the LDS-DMA rings, the persistent job queues and the asynchronous
operand prefetch are all timing-sensitive by construction, so each kernel is run repeatedly on the same inputs, with
other work in flight on a second stream, and every repetition must be bit-identical to the first
(the reference's own determinism check, tests/test_flash_attn.py:788-793, runs its backward 10 times)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def _noise(stream, n=8):
    """Unrelated bandwidth-heavy work on another stream, so that DMA latencies vary from run to run."""
    with torch.cuda.stream(stream):
        a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
        for _ in range(n):
            a = a @ a.t() * 1e-3
            a = a + 1.0
    return a


@pytest.mark.parametrize('shape', [(3, 1024, 16, 48, 768), (5, 640, 64, 16, 640), (2, 2048, 16, 48, 256), (9, 300, 4, 24, 104)])
def test_sense_mix_and_dc_repeat_bit_identically(shape):
    bp = _bp()
    b, s, k, dk, d = shape
    torch.manual_seed(5)
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
    c = torch.randn(b, s, k, d, device=DEV).bfloat16()
    dout = torch.randn(b, s, d, device=DEV).bfloat16()
    lse = bp.sense_lse(qk)
    first = bp.sense_mix(qk, c, lse=lse).clone()
    first_dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c).clone() if d % 8 == 0 and dk % 8 == 0 else None
    side = torch.cuda.Stream()
    for i in range(12):
        keep = _noise(side, 2 + i % 3)
        out = bp.sense_mix(qk, c, lse=lse)
        assert torch.equal(out, first), f'sense_mix repetition {i} differs'
        if first_dc is not None:
            dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c)
            assert torch.equal(dc, first_dc), f'sense_mix_dc repetition {i} differs'
    torch.cuda.synchronize()
    del keep


@pytest.mark.parametrize('cfg', [(4, 1024, 12, 64, True), (2, 1000, 8, 80, True), (3, 777, 6, 64, False), (2, 4096, 4, 64, True)])
def test_flash_fwd_repeats_bit_identically(cfg):
    bp = _bp()
    b, s, h, d, causal = cfg
    torch.manual_seed(6)
    qkv = torch.randn(b * s, 3, h, d, device=DEV).bfloat16()
    out = torch.empty_like(qkv[:, 0])
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
    lse0 = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, s, s, d ** -0.5, causal)[..., :s].clone()
    first = out.clone()
    side = torch.cuda.Stream()
    for i in range(12):
        keep = _noise(side, 2 + i % 3)
        out.fill_(float('nan'))
        lse = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, s, s, d ** -0.5, causal)
        # (LSE entries past the sequence length are padding the kernel never writes)
        assert torch.equal(out, first) and torch.equal(lse[..., :s], lse0), f'flash_fwd repetition {i} differs'
    torch.cuda.synchronize()
    del keep


def test_persistent_mix_launches_on_several_streams():
    """Persistent launches own one queue record each (a ring of 64 in the library): launches that overlap on different
    streams must neither share tickets nor wait on each other."""
    bp = _bp()
    torch.manual_seed(7)
    shapes = [(3, 512, 16, 48, 768), (2, 1024, 16, 48, 256), (4, 300, 4, 24, 104)]
    data = []
    for b, s, k, dk, d in shapes:
        qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
        c = torch.randn(b, s, k, d, device=DEV).bfloat16()
        lse = bp.sense_lse(qk)
        data.append((qk, c, lse, bp.sense_mix(qk, c, lse=lse).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in shapes]
    for rnd in range(6):
        outs = []
        for st, (qk, c, lse, _) in zip(streams, data):
            with torch.cuda.stream(st):
                for _ in range(3):   # several launches in flight per stream
                    out = bp.sense_mix(qk, c, lse=lse)
                outs.append(out)
        torch.cuda.synchronize()
        for out, (_, _, _, want) in zip(outs, data):
            assert torch.equal(out, want), f'round {rnd}: a concurrent launch produced a different result'


def test_persistent_mix_graphs_and_eager_launches_interleaved():
    """Two captured HIP graphs and eager launches on three streams, 200 interleaved replays: every launch owns its
    64-byte ticket record (`queue_ws`: the binding allocates one per call, so a graph replays its own), a memset node
    re-arms it in front of the kernel, and no two launches can ever share tickets -- all results bit-identical."""
    bp = _bp()
    torch.manual_seed(8)
    shapes = [(3, 512, 16, 48, 768), (2, 1024, 16, 48, 256), (4, 300, 4, 24, 104)]
    data = []
    for b, s, k, dk, d in shapes:
        qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
        c = torch.randn(b, s, k, d, device=DEV).bfloat16()
        dout = torch.randn(b, s, d, device=DEV).bfloat16()
        lse = bp.sense_lse(qk)
        data.append((qk, c, dout, lse, bp.sense_mix(qk, c, lse=lse).clone(),
                     bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c).clone(), dk))
    torch.cuda.synchronize()
    graphs = []
    for qk, c, dout, lse, _, _, dk in data[:2]:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bp.sense_mix(qk, c, lse=lse)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = bp.sense_mix(qk, c, lse=lse)
            dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c)
        graphs.append((g, out, dc))
    streams = [torch.cuda.Stream() for _ in range(3)]
    for rnd in range(200):
        outs = []
        for i, st in enumerate(streams):
            qk, c, dout, lse, _, _, dk = data[(i + rnd) % 3]
            with torch.cuda.stream(st):
                outs.append(((i + rnd) % 3, bp.sense_mix(qk, c, lse=lse), bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c)))
        for g, _, _ in graphs:
            g.replay()                       # on the current stream, concurrent with the three side streams
        if rnd % 20 == 19:
            torch.cuda.synchronize()
            for j, out, dc in outs:
                assert torch.equal(out, data[j][4]) and torch.equal(dc, data[j][5]), f'eager launch, round {rnd}'
            for (g, out, dc), d in zip(graphs, data):
                assert torch.equal(out, d[4]) and torch.equal(dc, d[5]), f'graph replay, round {rnd}'
    torch.cuda.synchronize()


def test_raw_abi_mix_with_and_without_queue_ws():
    """C ABI directly: queue_ws = NULL (the library's ring) and a caller-owned, deliberately dirty record give the
    same bits."""
    import ctypes
    bp = _bp()
    torch.manual_seed(9)
    b, s, k, dk, d = 2, 512, 16, 48, 768
    qk = torch.randn(b, s, 2, k, dk, device=DEV).bfloat16()
    c = torch.randn(b, s, k, d, device=DEV).bfloat16()
    want = bp.sense_mix(qk, c)
    ws = torch.empty(b, k, s, dtype=torch.float32, device=DEV)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for queue in (None, torch.full((16,), 0x7fffffff, dtype=torch.int32, device=DEV)):
        out = torch.zeros(b, s, d, device=DEV, dtype=torch.bfloat16)
        for _ in range(3):       # the dirty record is re-armed by every launch
            rc = bp.lib().bp_sense_mix(qk.data_ptr(), c.data_ptr(), out.data_ptr(), ws.data_ptr(), 0, b, s, k, dk, d,
                                       qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3), c.stride(0), c.stride(1),
                                       c.stride(2), out.stride(0), out.stride(1), dk ** -0.5, 1,
                                       queue.data_ptr() if queue is not None else None, stream)
            assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_null_queue_ws_is_refused_under_graph_capture():
    """C ABI 4: a persistent launch with queue_ws == NULL on a stream that is being captured returns BP_ERR_QUEUE_WS (-8)
    instead of capturing a graph that would replay on the library's shared record (round-3 review); with a caller-owned
    record the same capture succeeds and replays to the eager result.  Covers bp_sense_mix and bp_sense_mix_dc."""
    import ctypes
    bp = _bp()
    torch.manual_seed(10)
    b, s, k, dk, d = 1, 256, 4, 16, 256
    qk = torch.randn(b, s, 2, k, dk, device=DEV).bfloat16()
    c = torch.randn(b, s, k, d, device=DEV).bfloat16()
    dout = torch.randn(b, s, d, device=DEV).bfloat16()
    lse = bp.sense_lse(qk)
    want = bp.sense_mix(qk, c, lse=lse)
    want_dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c)
    out = torch.zeros(b, s, d, device=DEV, dtype=torch.bfloat16)
    dc = torch.zeros_like(c)
    queue = torch.empty(16, dtype=torch.int32, device=DEV)
    queue2 = torch.empty(16, dtype=torch.int32, device=DEV)

    def mix(q):
        return bp.lib().bp_sense_mix(qk.data_ptr(), c.data_ptr(), out.data_ptr(), lse.data_ptr(), 1, b, s, k, dk, d,
                                     qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3), c.stride(0), c.stride(1),
                                     c.stride(2), out.stride(0), out.stride(1), dk ** -0.5, 1, q,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def mix_dc(q):
        return bp.lib().bp_sense_mix_dc(qk.data_ptr(), dout.data_ptr(), lse.data_ptr(), dc.data_ptr(), b, s, k, dk, d,
                                        qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3), dout.stride(0),
                                        dout.stride(1), dc.stride(0), dc.stride(1), dc.stride(2), dk ** -0.5, 1, q,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    assert mix(None) == 0 and mix_dc(None) == 0          # eager launches may use the library's ring
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    codes = []
    with torch.cuda.graph(graph):
        codes += [mix(None), mix_dc(None)]                # refused: nothing is captured for these two
        codes += [mix(queue.data_ptr()), mix_dc(queue2.data_ptr())]
    assert codes == [-8, -8, 0, 0], codes
    assert b'queue_ws' in bp.lib().bp_strerror(-8)
    out.zero_(); dc.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(dc, want_dc)


def test_gathering_mix_repeats_bit_identically_and_replays_from_a_graph():
    """bp_sense_mix_gather keeps a per-job table of row offsets in LDS behind the DMA ring: repeated launches with other
    work in flight, launches on two streams at once and a captured graph all give the bits of bp_sense_mix on the gathered
    rows."""
    bp = _bp()
    torch.manual_seed(12)
    b, s, k, dk, d, rows = 3, 1024, 16, 48, 768, 4099
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
    table = torch.randn(rows, k, d, device=DEV).bfloat16()
    index = torch.randint(0, rows, (b, s), device=DEV, dtype=torch.int32)
    lse = bp.sense_lse(qk)
    want = bp.sense_mix(qk, table[index.long()], lse=lse).clone()
    side = torch.cuda.Stream()
    for i in range(10):
        keep = _noise(side, 2 + i % 3)
        assert torch.equal(bp.sense_mix_gather(qk, table, index, lse=lse), want), f'repetition {i} differs'
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for st in streams:
        with torch.cuda.stream(st):
            for _ in range(3):
                out = bp.sense_mix_gather(qk, table, index, lse=lse)
            outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs)
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        bp.sense_mix_gather(qk, table, index, lse=lse)
    torch.cuda.current_stream().wait_stream(warm)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = bp.sense_mix_gather(qk, table, index, lse=lse)
    for _ in range(5):
        out_g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_g, want)
    del keep
