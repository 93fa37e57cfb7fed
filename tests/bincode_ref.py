"""Test infrastructure: an independent, schema-driven Python restatement of the bincode 1.3.3 (default options) layout of
the reference's `MpnWork` and friends, written from the Rust type definitions - NOT from bazuka_amd/csrc/host_bincode.h:

  MpnWork / MpnConfig / ZkPublicInputs / MpnWorkData / *Transition   /root/reference/src/mpn/mod.rs:202-270, 426-511
  MpnAccount, MpnTransaction, ZkCompressedState, ZkVerifierKey, ZkProof  src/zk/mod.rs:59-65, 542-593, 566-571, 646-651
  ContractId, Money, ContractDeposit, ContractWithdraw, MpnDeposit, MpnWithdraw  src/core/transaction.rs:60-174
  jubjub PointAffine / PointCompressed / PublicKey / Signature          src/crypto/jubjub/curve.rs:10-14, mod.rs:32-52
  Groth16VerifyingKey / Groth16Proof                                    src/zk/groth16/mod.rs:19-38
  GetMpnWorkRequest/Response, PostMpnSolutionRequest/Response           src/client/messages.rs:368-386

bincode default options: little-endian fixed-width integers, u64 lengths for Vec / String / HashMap / byte strings, u32
enum variant index, Option = one tag byte, struct / tuple / fixed array / PhantomData without framing.  Values are plain
Python: ints, bools, bytes (fixed-size blobs), str, lists, dicts (struct fields / HashMap), (variant, payload) for enums.
"""
import struct

# ---- schema combinators: each is (encode(value) -> bytes, decode(buf, pos) -> (value, pos))


class T:
    def __init__(self, enc, dec):
        self.enc, self.dec = enc, dec


def _int(fmt, size):
    return T(lambda v: struct.pack(fmt, v), lambda b, p: (struct.unpack_from(fmt, b, _chk(b, p, size))[0], p + size))


def _chk(b, p, size):
    if p + size > len(b):
        raise ValueError(f"unexpected end of input at {p} (+{size})")
    return p


U8, U32, U64 = _int("<B", 1), _int("<I", 4), _int("<Q", 8)


def _bool_dec(b, p):
    v = b[_chk(b, p, 1)]
    if v > 1:
        raise ValueError("invalid bool")
    return bool(v), p + 1


BOOL = T(lambda v: b"\x01" if v else b"\x00", _bool_dec)


def Blob(n):  # fixed-size opaque bytes: tuples / arrays of integers carry no framing
    def enc(v):
        assert len(v) == n, (len(v), n)
        return bytes(v)
    return T(enc, lambda b, p: (bytes(b[_chk(b, p, n):p + n]), p + n))


def Struct(*fields):
    def enc(v):
        return b"".join(t.enc(v[name]) for name, t in fields)

    def dec(b, p):
        out = {}
        for name, t in fields:
            out[name], p = t.dec(b, p)
        return out, p
    return T(enc, dec)


def Vec(t):
    def enc(v):
        return U64.enc(len(v)) + b"".join(t.enc(x) for x in v)

    def dec(b, p):
        n, p = U64.dec(b, p)
        if n > len(b):
            raise ValueError("sequence length exceeds input")
        out = []
        for _ in range(n):
            x, p = t.dec(b, p)
            out.append(x)
        return out, p
    return T(enc, dec)


def Map(kt, vt):  # HashMap: entry order is unspecified on the wire; dict insertion order here
    def enc(v):
        return U64.enc(len(v)) + b"".join(kt.enc(k) + vt.enc(x) for k, x in v.items())

    def dec(b, p):
        n, p = U64.dec(b, p)
        if n > len(b):
            raise ValueError("map length exceeds input")
        out = {}
        for _ in range(n):
            k, p = kt.dec(b, p)
            out[k], p = vt.dec(b, p)
        return out, p
    return T(enc, dec)


def Enum(*variants):  # variants: (name, payload type or None); value = (name, payload)
    names = [n for n, _ in variants]

    def enc(v):
        name, payload = v
        i = names.index(name)
        t = variants[i][1]
        return U32.enc(i) + (t.enc(payload) if t else b"")

    def dec(b, p):
        i, p = U32.dec(b, p)
        if i >= len(variants):
            raise ValueError(f"enum variant {i}")
        name, t = variants[i]
        if t is None:
            return (name, None), p
        payload, p = t.dec(b, p)
        return (name, payload), p
    return T(enc, dec)


def Option(t):
    def enc(v):
        return b"\x00" if v is None else b"\x01" + t.enc(v)

    def dec(b, p):
        tag = b[_chk(b, p, 1)]
        if tag > 1:
            raise ValueError("invalid Option tag")
        if tag == 0:
            return None, p + 1
        return t.dec(b, p + 1)
    return T(enc, dec)


STRING = T(lambda v: U64.enc(len(v.encode())) + v.encode(),
           lambda b, p: (lambda n, q: (bytes(b[_chk(b, q, n):q + n]).decode(), q + n))(*U64.dec(b, p)))
BYTES = T(lambda v: U64.enc(len(v)) + bytes(v),
          lambda b, p: (lambda n, q: (bytes(b[_chk(b, q, n):q + n]), q + n))(*U64.dec(b, p)))

# ---- the reference's types
ZkScalar = Blob(32)                                   # struct ZkScalar([u64; 4]): Montgomery limbs
Fp = Blob(48)                                         # struct Fp([u64; 6])
G1 = Blob(97)                                         # (Fp, Fp, bool)
G2 = Blob(193)                                        # ((Fp, Fp), (Fp, Fp), bool)
PointAffine = Struct(("x", ZkScalar), ("y", ZkScalar))
PointCompressed = Struct(("x", ZkScalar), ("odd", BOOL))
ZkPublicKey = PointCompressed                         # jubjub::PublicKey(pub PointCompressed)
ZkSignature = Struct(("r", PointAffine), ("s", ZkScalar))
ContractId = Enum(("Null", None), ("Ziesha", None), ("Custom", ZkScalar))   # Null(PhantomData): no payload bytes
Money = Struct(("token_id", ContractId), ("amount", U64))
L1PublicKey = BYTES                                   # ed25519_dalek::PublicKey: serialize_bytes [recalled]
L1Signature = Blob(64)                                # ed25519::Signature (>= 1.3): 64-tuple [recalled]
L1SignatureLenPrefixed = BYTES


def contract_deposit(sig_t=L1Signature):
    return Struct(("memo", STRING), ("contract_id", ContractId), ("deposit_circuit_id", U32), ("calldata", ZkScalar),
                  ("src", L1PublicKey), ("amount", Money), ("fee", Money), ("nonce", U32), ("sig", Option(sig_t)))


ContractDeposit = contract_deposit()
ContractWithdraw = Struct(("memo", STRING), ("contract_id", ContractId), ("withdraw_circuit_id", U32), ("calldata", ZkScalar),
                          ("dst", L1PublicKey), ("amount", Money), ("fee", Money))
MpnWithdraw = Struct(("mpn_address", ZkPublicKey), ("mpn_withdraw_nonce", U32), ("mpn_sig", ZkSignature), ("payment", ContractWithdraw))
MpnAccount = Struct(("tx_nonce", U32), ("withdraw_nonce", U32), ("address", PointAffine), ("tokens", Map(U64, Money)))
MpnTransaction = Struct(("nonce", U32), ("src_pub_key", ZkPublicKey), ("dst_pub_key", ZkPublicKey), ("amount", Money), ("fee", Money),
                        ("sig", ZkSignature))
Proof = Vec(Blob(96))                                 # Vec<[ZkScalar; 3]>


def deposit_transition(sig_t=L1Signature):
    mpn_deposit = Struct(("mpn_address", ZkPublicKey), ("payment", contract_deposit(sig_t)))
    return Struct(("enabled", BOOL), ("tx", mpn_deposit), ("before", MpnAccount), ("before_balances_hash", ZkScalar),
                  ("before_balance", Money), ("proof", Proof), ("account_index", U64), ("token_index", U64), ("balance_proof", Proof))


WithdrawTransition = Struct(("enabled", BOOL), ("tx", MpnWithdraw), ("before", MpnAccount), ("before_token_balance", Money),
                            ("before_fee_balance", Money), ("proof", Proof), ("account_index", U64), ("token_index", U64),
                            ("token_balance_proof", Proof), ("before_token_hash", ZkScalar), ("fee_token_index", U64),
                            ("fee_balance_proof", Proof))
UpdateTransition = Struct(("enabled", BOOL), ("tx", MpnTransaction), ("src_before", MpnAccount), ("src_before_balances_hash", ZkScalar),
                          ("src_before_balance", Money), ("src_before_fee_balance", Money), ("src_proof", Proof), ("src_index", U64),
                          ("src_token_index", U64), ("src_balance_proof", Proof), ("src_fee_token_index", U64),
                          ("src_fee_balance_proof", Proof), ("dst_before", MpnAccount), ("dst_before_balances_hash", ZkScalar),
                          ("dst_before_balance", Money), ("dst_proof", Proof), ("dst_index", U64), ("dst_token_index", U64),
                          ("dst_balance_proof", Proof))
Groth16VerifyingKey = Struct(("alpha_g1", G1), ("beta_g1", G1), ("beta_g2", G2), ("gamma_g2", G2), ("delta_g1", G1), ("delta_g2", G2),
                             ("ic", Vec(G1)))
ZkVerifierKey = Enum(("Groth16", Groth16VerifyingKey))
Groth16Proof = Struct(("a", G1), ("b", G2), ("c", G1))
ZkProof = Enum(("Groth16", Groth16Proof))
MpnConfig = Struct(("log4_tree_size", U8), ("log4_token_tree_size", U8), ("log4_deposit_batch_size", U8), ("log4_withdraw_batch_size", U8),
                   ("log4_update_batch_size", U8), ("mpn_contract_id", ContractId), ("mpn_num_update_batches", U64),
                   ("mpn_num_deposit_batches", U64), ("mpn_num_withdraw_batches", U64), ("deposit_vk", ZkVerifierKey),
                   ("withdraw_vk", ZkVerifierKey), ("update_vk", ZkVerifierKey))
ZkPublicInputs = Struct(("height", U64), ("state", ZkScalar), ("aux_data", ZkScalar), ("next_state", ZkScalar))
ZkCompressedState = Struct(("state_hash", ZkScalar), ("state_size", U64))


def mpn_work(sig_t=L1Signature):
    data = Enum(("Deposit", Vec(deposit_transition(sig_t))), ("Withdraw", Vec(WithdrawTransition)), ("Update", Vec(UpdateTransition)))
    return Struct(("config", MpnConfig), ("public_inputs", ZkPublicInputs), ("data", data), ("new_root", ZkCompressedState),
                  ("reward", U64))


MpnWork = mpn_work()
Address = L1PublicKey
GetMpnWorkRequest = Struct(("address", Address))
GetMpnWorkResponse = Struct(("works", Map(U64, MpnWork)))
PostMpnSolutionRequest = Struct(("prover", Address), ("proofs", Map(U64, ZkProof)))
PostMpnSolutionResponse = Struct(("accepted", U64))
PostMpnWorkerRequest = Struct(("address", Address))
PostMpnWorkerResponse = Struct(("accepted", BOOL))


def encode(t, v) -> bytes:
    return t.enc(v)


def decode(t, b: bytes, exact=True):
    v, p = t.dec(b, 0)
    if exact and p != len(b):
        raise ValueError(f"{len(b) - p} trailing bytes")
    return v


def decode_prefix(t, b: bytes):
    return t.dec(b, 0)
