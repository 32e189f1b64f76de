"""scratch: gpurun_out/pmc_<tag>.json (scratch/pmc_run2.sh) -> profiles/{tag}_gemm_traffic.json (what bench.py's roofline.traffic reads)
and profiles/{tag}_gemm_pmc_raw.json."""
import json, sys, shutil
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
d = json.load(open(f'gpurun_out/pmc_{tag}.json'))
def find(sub):
  # round 6: the trunk shapes run k_gemm_nt_bf16_p64<EPI> (64-wide whole-line super-stages); 'pers<..>' patterns match it first
  ks = [k for k in d if sub.replace('pers<', 'p64<') in k] or [k for k in d if sub in k]
  if len(ks) > 1:      # prefer the name that ENDS with the pattern (k_slab_reduce vs k_slab_reduce_batch)
    ks = [k for k in ks if k.rstrip().endswith(sub)] or ks
  assert len(ks) == 1, (sub, ks)
  return d[ks[0]]
def hbm(e, extra=()):
  rd = 2 * e['FETCH_SIZE'] * 1024 + sum(2 * x['FETCH_SIZE'] * 1024 * x.get('per', 1) for x in extra)
  wr = e['WRITE_SIZE'] * 1024 + sum(x['WRITE_SIZE'] * 1024 * x.get('per', 1) for x in extra)
  return {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "avg_us_profiled": round(e.get('avg_us_profiled', float('nan')), 1),
          "mfma_busy_frac": round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024), 4) if 'GRBM_GUI_ACTIVE' in e else None}
red = dict(find('k_slab_reduce'), per=2)      # a TN call = the GEMM + two slab reductions (weights, bias)
out = {
  "nt_fwd": dict(hbm(find('pers<35>')), kernel='k_gemm_nt_bf16_p64<35> (pers<35> before round 6)'),
  "nt_dx": dict(hbm(find('pers<16>')), kernel='k_gemm_nt_bf16_p64<16>'),
  "nt_fwd_nobits": dict(hbm(find('pers<3>')), kernel='k_gemm_nt_bf16_p64<3>'),
  "nt_dx_bf16mask": dict(hbm(find('pers<4>')), kernel='k_gemm_nt_bf16_p64<4>'),
  "tn_dw": dict(hbm(find('k_gemm_tn_bf16_big'), [red]), kernel='k_gemm_tn_bf16_big + 2 x k_slab_reduce'),
  **({"tn_dw_batch": dict(hbm(find('k_gemm_tn_bf16_batch'), [dict(find('k_slab_reduce_batch'), per=1)]),
                          kernel='k_gemm_tn_bf16_batch (9 items = 8.5 trunk layer-equivalents, 2 pieces per tile) + k_slab_reduce_batch',
                          algorithmic_bytes=2 * 131072 * (7 * 1024 + 2 * 512 + 9 * 1024) + 4 * (7 * 1024 + 2 * 512) * 1024, flops=2.0 * 131072 * 1024 * (7 * 1024 + 2 * 512))}
     if any('tn_bf16_batch' in k for k in d) else {}),
  "_mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
  "_source": "scratch/pmc_run2.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 4 launches each, trunk shape "
             "M=131072 N=K=1024); HBM bytes = 2 x FETCH_SIZE(KB) x 1024 (gfx950 64-B request correction, MI355X_MICROARCH.md) "
             "+ WRITE_SIZE(KB) x 1024",
}
json.dump(out, open(f'profiles/{tag}_gemm_traffic.json', 'w'), indent=1)
shutil.copy(f'gpurun_out/pmc_{tag}.json', f'profiles/{tag}_gemm_pmc_raw.json')
print(json.dumps(out, indent=1))
