import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and "ms_pile" in v: print(k, "keys", v["ms_keys"], "sort", v["ms_sort"], "gather", v["ms_gather"], "pile", v["ms_pile"], "pairs", v["ms_probe_pairs"], "probe", v["ms_probe"], "total", v["ms_total"], "deferred", v["deferred_sources"], "own", v.get("pile_own_lists"))
