cd $GRAFT_REPO_ROOT
python tools/probe_bench.py --mix-reads 10000000 --steps 4 --rounds 2 --env DBTK_IDX_SPARSITY=2 DBTK_IDX_SPARSITY=2,DBTK_MZ_SPARSITY=3 DBTK_IDX_SPARSITY=2,DBTK_MZ_SPARSITY=3,DBTK_OVF_SPARSITY=8,DBTK_CLS_SPARSITY_PCT=130 > gpurun_out/fp1.log 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/fp1.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["env"], "allhit", d["all_hit"]["kernels"], d["all_hit"]["same_counts_as_first"]); print("    headline", d["headline"]["kernels"], d["headline"]["same_counts_as_first"])
    else: print(l[:300])
PY
for e in "" "DBTK_IDX_SPARSITY=2 DBTK_MZ_SPARSITY=3 DBTK_OVF_SPARSITY=8 DBTK_CLS_SPARSITY_PCT=130 DBTK_GRMZ_SPARSITY=3"; do echo "walk env: $e"; env $e python tools/walk_bench.py 80000 2000000 2>&1 | grep k_walk; done
