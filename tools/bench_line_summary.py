import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["value"])
r=d["roofline"]; print({k:r.get(k) for k in ("kernel","bound","achieved","frac","avg_us","traffic","mixed_roofline_frac","share_of_instrumented_step","us_per_step","launches_per_step")})
rs=d.get("roofline_symbol")
if rs: print("symbol:", {k:rs.get(k) for k in ("kernel","bound","achieved","frac","avg_us","traffic")})
e=d["extra"]
for t in e["top3"]: print((t["kernel"][:50], t["us_per_step"], t["launches_per_step"], t["bound"], t["frac"], t["mixed_roofline_frac"], t["pmc_traffic_over_algorithmic"]))
print(e["launch_tax"]); print(e.get("step_survey_frac"), e.get("step_survey_bound_us"), e.get("fp32_step"))
print(e["encoder_fwd"]["t_us"], {k:(v["us"],v["TFLOPs"]) for k,v in e["encoder_fwd"]["per_layer"].items()})
print(e.get("contrastive_4096x128"))
f=d["roofline"]
if "others" in f: [print("   ", (o["family"], o["us_per_step"], o["bound"], o["frac"], o["mixed_roofline_frac"])) for o in f["others"]]
