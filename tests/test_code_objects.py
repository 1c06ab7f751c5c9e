"""CPU: register account of the built gfx950 code objects (no GPU needed: hipcc cross-compiles, the metadata note of
every kernel carries its register / spill / scratch numbers).  The hot instantiations -- the ones BASELINE.json's
configs launch -- must not touch scratch memory: a scratch reload inside an LDS-DMA ring loop is a VMEM load whose
`s_waitcnt vmcnt(0)` also drains the tiles the ring just put in flight (compiler findings), and the
round-3 review found such spills in four shipped kernels.  scripts/kernel_resources.py prints the whole table."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import kernel_resources as KR  # noqa: E402

# (object, kernel name as kernel_resources prints it, waves per SIMD the registers must still allow[, spilled registers
# tolerated -- default 0])
HOT = [
    # Backpack-Small trunk attention (d_h = 64) and its LSE-only twin for the senses (d_k = 48), bf16 and fp16
    ('flash_fwd_dma.o', 'flash_fwd_dma_kernel<BF16, 4, 2, true, true, false>', 4),
    ('flash_fwd_dma.o', 'flash_fwd_dma_kernel<F16, 4, 2, true, true, false>', 4),
    ('flash_fwd_dma.o', 'flash_fwd_dma_kernel<BF16, 3, 2, false, false, false>', 4),
    # Mini (d_h = 80, d_k = 10 carried as 16)
    ('flash_fwd_dma.o', 'flash_fwd_dma_kernel<BF16, 5, 3, true, false, false>', 3),
    ('flash_fwd_dma.o', 'flash_fwd_dma_kernel<BF16, 1, 1, false, false, false>', 4),
    # fused sense mix: Small (d_k = 48) and Mini (16), 768 / 640 output columns
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<BF16, 3, true, false, false>', 2),
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<F16, 3, true, false, false>', 2),
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<BF16, 1, false, false, false>', 2),
    # the same with the content rows gathered from the per-token table (inference, bp_sense_mix_gather)
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<BF16, 3, true, false, true>', 2),
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<F16, 3, true, false, true>', 2),       # config 5 (S = 4096 fp16)
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<BF16, 1, false, false, true>', 2),     # config 4 (Mini k = 64: 640 columns)
    ('sense_mix_dma.o', 'sense_mix_dma_kernel<BF16, 2, false, false, true>', 2),     # config 1 (Micro: d_k = 24, 384 columns)
    # training step (config 3): attention backward at d_h = 64 with and without dropout, sense-mix dC
    # dK/dV at three waves per SIMD (168 registers): ONE 64-bit value still goes to scratch, stored in front of the
    # clean-tile loop and reloaded behind it, once per pass -- no tile loop of the kernel touches scratch (round 3: 12
    # spilled registers, one reload per edge step in front of the statistics DMA; scripts/kernel_resources.py)
    ('flash_bwd.o', 'flash_bwd_dkdv_kernel<BF16, 4, true, false>', 3, 2),
    ('flash_bwd.o', 'flash_bwd_dq_kernel<BF16, 4, true, false>', 3),
    ('flash_bwd.o', 'flash_bwd_dkdv_kernel<BF16, 4, true, true>', 2),
    ('flash_bwd.o', 'flash_bwd_dq_kernel<BF16, 4, true, true>', 2),
    ('sense_mix_bwd.o', 'sense_mix_dc_kernel<BF16, 3, true>', 2),
    # wide senses (the reference's vecs-4 / vecs-1 ablations): d_k <= 192 at two workgroups per CU; d_k <= 640 keeps its 160
    # fragment registers in the unified file (one wave per SIMD) -- neither may touch scratch
    ('sense_wide.o', 'sense_mix_wide_kernel<BF16, true, true, 12>', 2),
    ('sense_wide.o', 'sense_mix_wide_kernel<BF16, true, true, 40>', 1),
    ('sense_wide.o', 'sense_lse_wide_kernel<BF16, true, 12>', 3),
    ('sense_wide.o', 'sense_lse_wide_kernel<BF16, true, 40>', 1),
    # the same two configurations on the LDS-DMA ring (sense_wide_dma.hip): d_k = 160 as eight waves x 320 columns at two
    # waves per SIMD, d_k = 640 as four waves x 320 columns owning the file
    ('sense_wide_dma.o', 'sense_mix_wide_dma_kernel<BF16, 10, 8, 10, 2, false, 1>', 2),
    ('sense_wide_dma.o', 'sense_mix_wide_dma_kernel<BF16, 40, 4, 10, 2, false, 1>', 1),
    ('sense_wide_dma.o', 'sense_mix_wide_dma_kernel<BF16, 10, 8, 10, 2, true, 1>', 2),     # table form (inference)
    ('sense_wide_dma.o', 'sense_mix_wide_dma_kernel<BF16, 40, 4, 10, 2, true, 1>', 1),
    ('sense_wide_dma.o', 'sense_lse_wide_dma_kernel<BF16, 10, 8, 2, 1>', 4),
    ('sense_wide_dma.o', 'sense_lse_wide_dma_kernel<BF16, 40, 4, 2, 1>', 2),
]


@pytest.fixture(scope='module')
def table():
    if not KR.tools_available():
        pytest.skip('llvm-objcopy / clang-offload-bundler / llvm-readelf not found under /opt/rocm')
    import __graft_entry__  # noqa: F401  (puts the package on sys.path)
    import importlib.util
    spec = importlib.util.spec_from_file_location('bp_build_hip', os.path.join(ROOT, 'backpacks-flash-attn_amd', 'build_hip.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()   # no-op when the objects are current
    objs = sorted({e[0] for e in HOT})
    ks = KR.kernels([os.path.join(KR.BUILD, o) for o in objs])
    return {(k['object'], k['name']): k for k in ks}


@pytest.mark.parametrize('entry', HOT, ids=[e[1] for e in HOT])
def test_hot_kernel_has_no_scratch(table, entry):
    obj, name, min_waves = entry[:3]
    tolerated = entry[3] if len(entry) > 3 else 0
    k = table.get((obj, name))
    assert k is not None, 'kernel not found in %s: %s' % (obj, name)
    assert k['vgpr_spill_count'] <= tolerated, k
    assert k['private_segment_fixed_size'] <= 4 * tolerated + (4 if tolerated else 0), k   # (+ the slot alignment)
    assert k['waves_per_simd'] >= min_waves, k


def test_resource_table_lists_every_object(table):
    assert len(table) > 50
    assert all('vgpr_count' in k and 'sgpr_spill_count' in k for k in table.values())


def test_hazard_scanner_flags_what_it_should():
    """scripts/mfma_hazard_scan.py on hand-made listings: the round-6 finding (a VALU read of an MFMA result one slot behind
    the loop's back-edge) and its cures are told apart; a transcendental's result read by the next plain VALU is flagged,
    forwarded to another transcendental or one instruction later it is not."""
    import mfma_hazard_scan as HS

    def listing(lines):
        return [(0x1000 + 4 * i, mn, ops) for i, (mn, ops) in enumerate(lines)]
    mfma = ('v_mfma_f32_32x32x16_bf16', 'v[0:15], v[66:69], v[44:47], v[0:15]')
    # block at 0x1000 is the loop head; the MFMA sits at index 3, the back-edge right behind it: offset = -(5 words)
    bad = listing([('v_max_f32_e32', 'v59, v3, v3'), ('s_nop', '7'), ('v_max3_f32', 'v65, v0, v1, v4'),
                   mfma, ('s_cbranch_scc1', str(65536 - 5)), ('s_endpgm', '')])
    assert [h[4].split()[0] for h in HS.scan('bad', bad)] == ['v_max_f32_e32']
    padded = listing([('v_max_f32_e32', 'v59, v3, v3'), mfma, ('s_nop', '7'), ('s_nop', '3'),
                      ('s_cbranch_scc1', str(65536 - 5)), ('s_endpgm', '')])
    assert HS.scan('padded', padded) == []
    chain = listing([mfma, mfma, ('s_nop', '7'), ('s_nop', '2'), ('v_add_f32_e32', 'v20, v0, v1'), ('s_endpgm', '')])
    assert HS.scan('chain', chain) == []            # the accumulator operand of the next MFMA is forwarded in hardware
    assert HS.passes('v_mfma_f32_32x32x16_bf16') == 8 and HS.passes('v_mfma_f32_16x16x32_f16') == 4
    trans_bad = listing([('v_exp_f32_e32', 'v1, v0'), ('v_add_f32_e32', 'v2, v1, v1'), ('s_endpgm', '')])
    assert len(HS.scan('t', trans_bad)) == 1
    trans_ok = listing([('v_exp_f32_e32', 'v1, v0'), ('v_exp_f32_e32', 'v3, v1'), ('v_mov_b32_e32', 'v9, v8'),
                        ('v_add_f32_e32', 'v2, v3, v3'), ('s_endpgm', '')])
    assert HS.scan('t', trans_ok) == []


def test_no_vector_instruction_reads_an_mfma_result_early(table):
    """gfx950 does not interlock "matrix pipe writes a VGPR -> VALU reads it"; hipcc pads the gap with s_nop, but not
    reliably across a branch (csrc/bp_common.h, settle_acc: found in round 6 as a one-ulp launch-to-launch variation of the
    wide LSE kernel).  scripts/mfma_hazard_scan.py walks every path behind every v_mfma of every kernel of the library;
    nothing may name a destination register of the MFMA before its wait states are over."""
    import mfma_hazard_scan as HS
    if not os.path.exists(os.path.join(KR.LLVM, 'llvm-objdump')):
        pytest.skip('llvm-objdump not found under /opt/rocm')
    hits = []
    import glob
    for obj in sorted(glob.glob(os.path.join(KR.BUILD, '*.o'))):
        for name, ins in HS.functions(HS.disassemble(obj)):
            hits += HS.scan(name, ins)
    assert not hits, hits[:5]


@pytest.mark.parametrize('obj', ['flash_fwd_dma.o', 'flash_bwd.o', 'sense_mix_dma.o', 'sense_mix_bwd.o', 'sense_wide_dma.o'])
def test_no_instruction_reads_m0_besides_the_lds_dma(obj):
    """csrc/bp_dma.h: `dma16_s` writes M0 and names it as clobbered instead of saving and restoring it (two scalar moves per
    1-KiB piece less).  That is only sound while no compiler-generated instruction in these kernels READS M0 -- relative
    register indexing (v_movrel / s_movrel / s_set_gpr_idx), s_sendmsg, v_interp -- so the disassembly of every code object
    that uses the helper is checked for them (advisor, round 3).  Every textual use of m0 must be an `s_mov_b32` of our own
    DMA statements: the write in front of the LDS-DMA instruction (which reads it implicitly), or the save / restore pair
    of the `dma16` / `dma4` forms."""
    import subprocess
    import tempfile
    if not KR.tools_available() or not os.path.exists(os.path.join(KR.LLVM, 'llvm-objdump')):
        pytest.skip('llvm tools not found under /opt/rocm')
    path = os.path.join(KR.BUILD, obj)
    if not os.path.exists(path):
        pytest.skip('objects not built')
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        KR._run(os.path.join(KR.LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, path)
        KR._run(os.path.join(KR.LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=' + KR.TARGET,
                '--input=' + fat, '--output=' + co)
        text = subprocess.run([os.path.join(KR.LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True,
                              text=True, check=True).stdout
    forbidden = ('v_movrel', 's_movrel', 's_set_gpr_idx', 's_sendmsg', 'v_interp_')
    m0_users = set()
    for line in text.splitlines():
        ops = line.split('//')[0].split()
        if not ops:
            continue
        assert not ops[0].startswith(forbidden), line
        if any(tok.strip(',') == 'm0' for tok in ops[1:]):
            m0_users.add(ops[0])
    assert m0_users <= {'s_mov_b32'}, m0_users
    assert 'global_load_lds_dwordx4' in text
