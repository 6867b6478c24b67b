import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if k.startswith("ops"):
        for o in v: print(o)
    elif k not in ("copies",): print(k, v)
