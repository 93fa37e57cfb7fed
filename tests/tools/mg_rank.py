"""One rank of a process-per-GPU device group (bzk_mg_create_rank), launched by tests/test_gpu_mg.py and by nothing else:
argv = rank world uid_hex n seed g2 exchange; prints the hex of the result every rank receives."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, uid, n, seed, g2, exchange = sys.argv[1:8]
    rank, world, n, seed, g2, exchange = int(rank), int(world), int(n), int(seed), int(g2), int(exchange)
    import torch
    from bazuka_amd import Bzk, Mg
    from util import rand_scalars_bytes, to_dev
    torch.cuda.set_device(0)
    mg = Mg(device=0, rank=rank, world=world, uid=bytes.fromhex(uid), exchange=exchange)
    ctx = Bzk(0)
    bases = torch.empty(n * (192 if g2 else 96), dtype=torch.uint8, device="cuda")
    (ctx.g2_synth_bases_dev if g2 else ctx.g1_synth_bases_dev)(seed, 0, n, bases)
    ctx.sync()
    sc = to_dev(rand_scalars_bytes(n, seed))
    torch.cuda.synchronize()
    hb = mg.bases_load_dev([bases], n, g2=bool(g2))
    outs, errs = [], 0
    for _ in range(4 if os.environ.get("BZK_MG_TEST_FAULT") else 3):  # several calls: the double-buffered exchange is re-used
        try:
            outs.append(mg.msm_dev(hb, [sc], n, g2=bool(g2)))
        except Exception as e:   # an injected fault on ANY rank must surface on EVERY rank, for that call only
            errs += 1
            print("ERROR", rank, str(e).replace("\n", " "), flush=True)
    assert all(o == outs[0] for o in outs)
    print("RESULT", rank, mg.exchange, outs[0].hex(), len(outs), errs, flush=True)
    mg.bases_free(hb)
    mg.close()
    ctx.close()


if __name__ == "__main__":
    main()
