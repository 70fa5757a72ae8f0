import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print(sys.argv[2], "step %.4f lines %.4f" % (j["ms_per_step"], j["kernel_ms_per_step"]["lines"]), {k:(round(r[k],4) if isinstance(r.get(k),float) else r.get(k)) for k in ("valu_busy","valu_insts_per_launch","salu_per_valu","f64_share_of_valu_insts","traffic")})
