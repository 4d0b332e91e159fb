import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], round(d["ms_per_step"],4), "tile", round(d["roofline"]["kernel_ms_per_step"],4))
for p in d["passes"]: print("   k",p["k"],"act",p["active"],"scr",p["screened"],"H",p["H_formed"],"tile_ms",p["tile_ms"])
