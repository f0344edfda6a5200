"""Per kernel: instructions per wave by class and the share of the SIMD's vector-pipe time that MFMAs can occupy AT BEST under
the additive cost model measured by tools/ablate/pipe_overlap.hip (profiles/r11e_pipe_overlap.txt): on gfx950 an fp32 MFMA
(16x16x4: 32 cycles) holds its SIMD like a long VALU instruction -- other VALU work and LDS returns, of the same wave or of the
co-resident one, ADD to it instead of hiding under it.  Costs used: 32 cycles per fp32 MFMA (64 for 32x32x2), 2.8 per other VALU
instruction (v_fma-class at two waves per SIMD; v_pk_* 5.0), 16 per LDS instruction (ds_read_b128 at 64 B/clk/CU).
usage: python tools/instruction_mix.py <pmc_summary.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))


def f(r, k):
    try:
        return float(r.get(k) or 0.0)
    except ValueError:
        return 0.0


probe = [r for r in rows if 'mfma_probe' in r['kernel']]
incl = None
if probe:
    p = probe[0]
    incl = f(p, 'SQ_INSTS_VALU') >= 0.9 * f(p, 'SQ_INSTS_MFMA')
    print(f"calibration (mfma_probe_kernel): SQ_INSTS_VALU {f(p, 'SQ_INSTS_VALU'):.0f}, SQ_INSTS_MFMA {f(p, 'SQ_INSTS_MFMA'):.0f} "
          f"-> SQ_INSTS_VALU {'INCLUDES' if incl else 'does NOT include'} the MFMAs")
print(f"{'kernel':58s} {'waves':>7s} {'mfma/w':>7s} {'valu/w':>7s} {'lds/w':>6s} {'vmem/w':>6s} {'valu per mfma':>13s} {'mfma share of vector-pipe time (model)':>38s}")
for r in sorted(rows, key=lambda r: -f(r, 'SQ_INSTS_MFMA')):
    waves = f(r, 'SQ_WAVES')
    mf = f(r, 'SQ_INSTS_MFMA')
    if waves <= 0 or mf <= 0:
        continue
    valu = f(r, 'SQ_INSTS_VALU') - (mf if incl else 0.0)
    lds, vmem = f(r, 'SQ_INSTS_LDS'), f(r, 'SQ_INSTS_VMEM')
    cyc = 64.0 if 'corr_gemm' in r['kernel'] else 32.0
    share = cyc * mf / (cyc * mf + 2.8 * valu + 16.0 * lds)
    print(f"{r['kernel'][:58]:58s} {waves:7.0f} {mf / waves:7.0f} {valu / waves:7.0f} {lds / waves:6.0f} {vmem / waves:6.0f} {valu / mf:13.2f} {share:38.3f}")
