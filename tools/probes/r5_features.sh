# Round 5: the keyword leg with the index features the timed corpus does not have — word-prefix databases (threshold 50),
# synonyms, phrases, negative terms (rb_prepare_queries_ex flags 15) — on the 10 M-document corpus, 256 callers
mkdir -p gpurun_out
MSI_SEARCH_CPU_PROFILE=1 timeout 85 python tools/kw_leg.py --callers 256 --queries 2304 --passes 2 --flags 15 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r5_features.json
cut -c1-1500 gpurun_out/r5_features.json
