import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"])
o=j["roofline"]["others"]
for k in sys.argv[2:]:
    print(k, o.get(k))
