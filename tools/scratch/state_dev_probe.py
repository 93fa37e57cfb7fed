import sys, time, json, random, faulthandler; faulthandler.enable()
sys.path.insert(0, '.')
from bazuka_amd import Bzk, DeviceState
from bench import _fr
ctx = Bzk(0)
def mb(m):
    if m[0] == "scalar": return (0).to_bytes(4, "little")
    if m[0] == "struct": return (1).to_bytes(4, "little") + len(m[1]).to_bytes(8, "little") + b"".join(mb(f) for f in m[1])
    return (2).to_bytes(4, "little") + bytes([m[1]]) + mb(m[2])
S_ = ("scalar",)
model = mb(("list", 15, ("struct", [S_, S_, S_, S_, ("list", 3, ("struct", [S_, S_]))])))
rnd = random.Random(5)
pairs = []
for a in rnd.sample(range(4 ** 15), 4096):
    for j in range(4): pairs.append(((a, j), _fr(rnd.randrange(1, 1 << 60))))
    pairs.append(((a, 4, 0, 0), _fr(1))); pairs.append(((a, 4, 0, 1), _fr(rnd.randrange(1, 1 << 40))))
dev = DeviceState(ctx, model)
dev.update(pairs, 1)
accts = sorted({p[0][0] for p in pairs})
res = {}
for name, n in (("update_1024", 512), ("update_16", 8), ("update_2", 1)):
    ts = []
    for rep in range(6):
        delta = []
        for a in rnd.sample(accts, n):
            delta.append(((a, 0), _fr(rnd.randrange(1, 1 << 30)))); delta.append(((a, 4, 0, 1), _fr(rnd.randrange(1, 1 << 40))))
        t = time.perf_counter(); dev.update(delta, 2 + rep); ts.append(time.perf_counter() - t)
    res[name] = [round(x * 1e3, 3) for x in ts]
for name, n in (("prove_64", 64), ("prove_1", 1)):
    ts = []
    for rep in range(6):
        t = time.perf_counter(); dev.prove((), accts[rep:rep + n]); ts.append(time.perf_counter() - t)
    res[name] = [round(x * 1e3, 3) for x in ts]
ts = []
for rep in range(6):
    locs = [(a, 0) for a in accts[:256]]
    t = time.perf_counter(); dev.get(locs); ts.append(time.perf_counter() - t)
res["get_256"] = [round(x * 1e3, 3) for x in ts]
ctx.prof_enable(True); ctx.prof_reset()
delta = []
for a in rnd.sample(accts, 512):
    delta.append(((a, 0), _fr(rnd.randrange(1, 1 << 30)))); delta.append(((a, 4, 0, 1), _fr(rnd.randrange(1, 1 << 40))))
dev.update(delta, 99)
res["kernels_update_1024"] = ctx.prof_dump()
print(json.dumps(res))
