import json,sys,subprocess,os
def run(env):
    e=dict(os.environ); e.update(env)
    r=subprocess.run([sys.executable,"bench.py","--no-cpu-baseline","--no-extras","--steps","5","--warmup","2"],env=e,capture_output=True,text=True)
    d=json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    l=d["latency_batch1_ms"]
    return l["p50"], {x["launch"]:x["us"] for x in l["per_launch_us"]}
a=run({}); b=run({"HNET_S3_TILE":"21"})
print("default p50",a[0],"tile21 p50",b[0])
for k in a[1]: print(f"{k:22s} {a[1][k]:7.2f} {b[1].get(k,0):7.2f}")
