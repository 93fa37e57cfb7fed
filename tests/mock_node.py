"""Test infrastructure: a stand-in for the three MPN-worker endpoints of a Bazuka node (/root/reference/src/node/mod.rs:393-413)
over plain HTTP on 127.0.0.1.  Requests are decoded with the reference-schema codec (tests/bincode_ref.py) and solutions
are judged the way `MpnWork::verify` does (src/mpn/mod.rs:281-295): Groth16 pairing check - the ORACLE's verifier - of
the proof against the work's verifying key and the public inputs [H(prover, reward), height, state, aux_data, next_state]."""
import hashlib
import threading
from http.server import BaseHTTPRequestHandler, HTTPServer

import bincode_ref as B
from oracle import pyref as pr


class MockNode:
    def __init__(self, works: dict, allowed_workers=None):
        """works: {work id: bincode MpnWork bytes}"""
        self.works = dict(works)
        self.allowed = allowed_workers
        self.solved = {}
        self.log = []
        node = self

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def _body(self):
                return self.rfile.read(int(self.headers.get("Content-Length", "0")))

            def _reply(self, data: bytes, status=200):
                self.send_response(status)
                self.send_header("Content-Length", str(len(data)))
                self.end_headers()
                self.wfile.write(data)

            def do_GET(self):
                body = self._body()
                if self.path != "/bincode/mpn/work":
                    return self._reply(b"", 404)
                addr = B.decode(B.GetMpnWorkRequest, body)["address"]
                node.log.append(("work", addr))
                if node.allowed is not None and addr not in node.allowed:
                    return self._reply(B.encode(B.U64, 0))
                out = [B.encode(B.U64, len(node.works) - len(node.solved))]
                for wid, blob in node.works.items():
                    if wid not in node.solved:
                        out += [B.encode(B.U64, wid), blob]
                self._reply(b"".join(out))

            def do_POST(self):
                body = self._body()
                if self.path == "/bincode/mpn/worker":
                    addr = B.decode(B.PostMpnWorkerRequest, body)["address"]
                    node.log.append(("worker", addr))
                    return self._reply(B.encode(B.PostMpnWorkerResponse, {"accepted": True}))
                if self.path != "/bincode/mpn/solution":
                    return self._reply(b"", 404)
                req = B.decode(B.PostMpnSolutionRequest, body)
                accepted = 0
                for wid, (variant, proof) in req["proofs"].items():
                    if wid in node.works and wid not in node.solved and variant == "Groth16" and node.verify(wid, req["prover"], proof):
                        node.solved[wid] = req["prover"]
                        accepted += 1
                node.log.append(("solution", req["prover"], sorted(req["proofs"]), accepted))
                self._reply(B.encode(B.PostMpnSolutionResponse, {"accepted": accepted}))

        self.httpd = HTTPServer(("127.0.0.1", 0), H)
        self.port = self.httpd.server_address[1]
        self.thread = threading.Thread(target=self.httpd.serve_forever, daemon=True)
        self.thread.start()

    def verify(self, wid, prover: bytes, proof: dict) -> bool:
        w = B.decode(B.MpnWork, self.works[wid])
        kind = ("Deposit", "Withdraw", "Update").index(w["data"][0])
        vk_val = w["config"][("deposit_vk", "withdraw_vk", "update_vk")[kind]][1]
        vk = pr.vk_from_bytes(B.encode(B.Groth16VerifyingKey, vk_val))
        pre = B.encode(B.Address, prover) + B.encode(B.U64, w["reward"])
        commitment = int.from_bytes(hashlib.sha3_256(pre).digest(), "little") % pr.R_MOD
        pi = w["public_inputs"]
        pub = [commitment, pi["height"]] + [pr.fr_from_mont_bytes(pi[k]) for k in ("state", "aux_data", "next_state")]
        try:
            return bool(pr.groth16_verify(vk, pub, pr.proof_from_bytes(proof["a"] + proof["b"] + proof["c"])))
        except Exception:
            return False

    def close(self):
        self.httpd.shutdown()
        self.httpd.server_close()
