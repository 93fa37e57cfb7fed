//! bzk-fixture-dump: fixed-rng vectors FROM THE REFERENCE for `tests/test_bellman_vectors_cpu.py`.
//!
//! **UNVERIFIED BY COMPILATION** in the image that builds libbzk (no rustc).  What it does, for each of the reference's three circuits at
//! the shape of its own tests - `UpdateCircuit` / `DepositCircuit` / `WithdrawCircuit` with L = 3, T = 3, log4 batch 1, four `::null`
//! transitions, commitment 456, height 0, state = next_state = 123 (`/root/reference/src/mpn/circuits/test.rs:117-149, 152-193, 196-237`):
//!
//!   1. synthesizes the circuit into a RECORDING constraint system that does exactly what bellman's `ProvingAssignment` does
//!      (bellman 0.14 `groth16/prover.rs`: `alloc` / `alloc_input` push the value, `enforce` evaluates the three linear combinations and
//!      marks the densities of A over aux and of B over inputs and aux; `create_proof` appends one `input_i * 0 = 0` row per input) and
//!      hashes what it saw: z = inputs ++ aux, A.z, B.z, C.z, the three density bit vectors;
//!   2. `generate_random_parameters` with `ChaCha20Rng::from_seed([7; 32])` -> sha256 of `Parameters::write`, the verifying key, the sizes
//!      of the five queries; with `--params-dir DIR` the three parameter files are kept (~70 MB each) so that the GPU test can load them
//!      through `bzk_params_load_bellman` and compare PROOF BYTES;
//!   3. r, s drawn from `ChaCha20Rng::from_seed([9; 32])` (r first, as `create_random_proof` does), `create_proof(circuit, &params, r, s)`,
//!      the proof checked with the reference's own `groth16_verify`-equivalent call, and A, B, C written uncompressed.
//!
//! Everything is written as hex / sha256 into ONE json (default: ../../tests/golden/bellman_vectors.json, ~6 KB).  Scalars are hashed as
//! their canonical 32-byte little-endian encoding (`Scalar::to_bytes`), points as `to_uncompressed()`; densities one byte (0 / 1) per
//! variable.  The consuming test converts libbzk's Montgomery limbs to the same encodings before hashing.
//!
//! usage:  cargo run --release -- [--out FILE] [--params-dir DIR]
use bazuka::core::{ContractId, Money, ZkHasher};
use bazuka::mpn::circuits::{DepositCircuit, UpdateCircuit, WithdrawCircuit};
use bazuka::mpn::{DepositTransition, UpdateTransition, WithdrawTransition};
use bazuka::zk::{ZkDataLocator, ZkDeltaPairs, ZkScalar, ZkStateBuilder, ZkStateModel};
use bellman::groth16::{self, Parameters};
use bellman::{Circuit, ConstraintSystem, Index, LinearCombination, SynthesisError, Variable};
use bls12_381::{Bls12, Scalar};
use ff::Field;
use rand_chacha::ChaCha20Rng;
use rand_core::SeedableRng;
use sha2::{Digest, Sha256};

const SETUP_SEED: [u8; 32] = [7u8; 32];
const PROOF_SEED: [u8; 32] = [9u8; 32];

/// bellman's `ProvingAssignment`, observable: same order of pushes, same density rules (prover.rs: `eval` marks a variable in the density
/// tracker when its coefficient is non-zero; A tracks aux only, B tracks inputs and aux, C nothing).
#[derive(Default)]
struct Recorder {
    inputs: Vec<Scalar>,
    aux: Vec<Scalar>,
    a: Vec<Scalar>,
    b: Vec<Scalar>,
    c: Vec<Scalar>,
    a_aux_density: Vec<u8>,
    b_input_density: Vec<u8>,
    b_aux_density: Vec<u8>,
}

impl Recorder {
    fn eval(&self, lc: &LinearCombination<Scalar>, input_density: Option<&mut Vec<u8>>, aux_density: Option<&mut Vec<u8>>) -> Scalar {
        let (mut idn, mut adn) = (input_density, aux_density);
        let mut acc = Scalar::zero();
        for (var, coeff) in lc.as_ref().iter() {
            let value = match var.get_unchecked() {
                Index::Input(i) => {
                    if let Some(d) = idn.as_deref_mut() {
                        if !bool::from(coeff.is_zero()) {
                            d[i] = 1;
                        }
                    }
                    self.inputs[i]
                }
                Index::Aux(i) => {
                    if let Some(d) = adn.as_deref_mut() {
                        if !bool::from(coeff.is_zero()) {
                            d[i] = 1;
                        }
                    }
                    self.aux[i]
                }
            };
            acc += value * coeff;
        }
        acc
    }
}

impl ConstraintSystem<Scalar> for Recorder {
    type Root = Self;
    fn alloc<F, A, AR>(&mut self, _: A, f: F) -> Result<Variable, SynthesisError>
    where
        F: FnOnce() -> Result<Scalar, SynthesisError>,
        A: FnOnce() -> AR,
        AR: Into<String>,
    {
        self.aux.push(f()?);
        self.a_aux_density.push(0);
        self.b_aux_density.push(0);
        Ok(Variable::new_unchecked(Index::Aux(self.aux.len() - 1)))
    }
    fn alloc_input<F, A, AR>(&mut self, _: A, f: F) -> Result<Variable, SynthesisError>
    where
        F: FnOnce() -> Result<Scalar, SynthesisError>,
        A: FnOnce() -> AR,
        AR: Into<String>,
    {
        self.inputs.push(f()?);
        self.b_input_density.push(0);
        Ok(Variable::new_unchecked(Index::Input(self.inputs.len() - 1)))
    }
    fn enforce<A, AR, LA, LB, LC>(&mut self, _: A, a: LA, b: LB, c: LC)
    where
        A: FnOnce() -> AR,
        AR: Into<String>,
        LA: FnOnce(LinearCombination<Scalar>) -> LinearCombination<Scalar>,
        LB: FnOnce(LinearCombination<Scalar>) -> LinearCombination<Scalar>,
        LC: FnOnce(LinearCombination<Scalar>) -> LinearCombination<Scalar>,
    {
        let (a, b, c) = (a(LinearCombination::zero()), b(LinearCombination::zero()), c(LinearCombination::zero()));
        // the densities are fields of self: take them out while `eval` borrows the assignments
        let mut ad = std::mem::take(&mut self.a_aux_density);
        let mut bi = std::mem::take(&mut self.b_input_density);
        let mut ba = std::mem::take(&mut self.b_aux_density);
        let av = self.eval(&a, None, Some(&mut ad));
        let bv = self.eval(&b, Some(&mut bi), Some(&mut ba));
        let cv = self.eval(&c, None, None);
        self.a_aux_density = ad;
        self.b_input_density = bi;
        self.b_aux_density = ba;
        self.a.push(av);
        self.b.push(bv);
        self.c.push(cv);
    }
    fn push_namespace<NR, N>(&mut self, _: N)
    where
        NR: Into<String>,
        N: FnOnce() -> NR,
    {
    }
    fn pop_namespace(&mut self) {}
    fn get_root(&mut self) -> &mut Self::Root {
        self
    }
}

fn sha_scalars(v: &[Scalar]) -> String {
    let mut h = Sha256::new();
    for s in v {
        h.update(s.to_bytes());
    }
    hex::encode(h.finalize())
}
fn sha_bytes(v: &[u8]) -> String {
    hex::encode(Sha256::digest(v))
}

/// the recording pass: what `create_proof` does before its MSMs (ONE input, synthesize, the trailing input rows)
fn record<C: Circuit<Scalar>>(circuit: C) -> Recorder {
    let mut cs = Recorder::default();
    cs.alloc_input(|| "", || Ok(Scalar::one())).unwrap();
    circuit.synthesize(&mut cs).expect("synthesize");
    for i in 0..cs.inputs.len() {
        cs.enforce(|| "", |lc| lc + Variable::new_unchecked(Index::Input(i)), |lc| lc, |lc| lc);
    }
    cs
}

fn dump<C: Circuit<Scalar> + Clone>(name: &str, circuit: C, public: &[ZkScalar], params_dir: Option<&str>) -> serde_json::Value {
    let rec = record(circuit.clone());
    let z: Vec<Scalar> = rec.inputs.iter().chain(rec.aux.iter()).cloned().collect();
    // --- setup with a fixed rng
    let params: Parameters<Bls12> = groth16::generate_random_parameters::<Bls12, _, _>(circuit.clone(), &mut ChaCha20Rng::from_seed(SETUP_SEED)).expect("setup");
    let mut blob = Vec::new();
    params.write(&mut blob).expect("Parameters::write");
    if let Some(dir) = params_dir {
        std::fs::write(format!("{}/{}.params", dir, name), &blob).expect("params file");
    }
    // --- proof with fixed blinding factors, drawn the way create_random_proof draws them (r, then s)
    let mut rng = ChaCha20Rng::from_seed(PROOF_SEED);
    let r = Scalar::random(&mut rng);
    let s = Scalar::random(&mut rng);
    let proof = groth16::create_proof(circuit, &params, r, s).expect("prove");
    let pvk = groth16::prepare_verifying_key(&params.vk);
    let inputs: Vec<Scalar> = public.iter().map(|x| (*x).into()).collect();
    assert!(groth16::verify_proof(&pvk, &proof, &inputs).is_ok(), "{}: the proof does not verify", name);
    assert_eq!(&rec.inputs[1..], &inputs[..], "{}: recorded public inputs differ from the circuit's fields", name);
    let g1s = |v: &[bls12_381::G1Affine]| {
        let mut h = Sha256::new();
        for p in v {
            h.update(p.to_uncompressed());
        }
        hex::encode(h.finalize())
    };
    let g2s = |v: &[bls12_381::G2Affine]| {
        let mut h = Sha256::new();
        for p in v {
            h.update(p.to_uncompressed());
        }
        hex::encode(h.finalize())
    };
    serde_json::json!({
        "circuit": name,
        "shape": {"log4_tree_size": 3, "log4_token_tree_size": 3, "log4_batch_size": 1},
        "public_inputs_le": public.iter().map(|x| hex::encode(Scalar::from(*x).to_bytes())).collect::<Vec<_>>(),
        "n_inputs": rec.inputs.len(), "n_aux": rec.aux.len(), "n_constraints": rec.a.len(),
        "sha256": {
            "z": sha_scalars(&z), "az": sha_scalars(&rec.a), "bz": sha_scalars(&rec.b), "cz": sha_scalars(&rec.c),
            "a_aux_density": sha_bytes(&rec.a_aux_density), "b_input_density": sha_bytes(&rec.b_input_density), "b_aux_density": sha_bytes(&rec.b_aux_density),
        },
        "density_totals": {"a_aux": rec.a_aux_density.iter().map(|&x| x as u64).sum::<u64>(),
                           "b_input": rec.b_input_density.iter().map(|&x| x as u64).sum::<u64>(),
                           "b_aux": rec.b_aux_density.iter().map(|&x| x as u64).sum::<u64>()},
        "setup": {
            "rng": "ChaCha20Rng::from_seed([7; 32])",
            "parameters_write_sha256": sha_bytes(&blob), "parameters_write_len": blob.len(),
            "query_len": {"h": params.h.len(), "l": params.l.len(), "a": params.a.len(), "b_g1": params.b_g1.len(), "b_g2": params.b_g2.len()},
            "query_sha256": {"h": g1s(&params.h), "l": g1s(&params.l), "a": g1s(&params.a), "b_g1": g1s(&params.b_g1), "b_g2": g2s(&params.b_g2)},
            "vk": {"alpha_g1": hex::encode(params.vk.alpha_g1.to_uncompressed()), "beta_g1": hex::encode(params.vk.beta_g1.to_uncompressed()),
                   "beta_g2": hex::encode(params.vk.beta_g2.to_uncompressed()), "gamma_g2": hex::encode(params.vk.gamma_g2.to_uncompressed()),
                   "delta_g1": hex::encode(params.vk.delta_g1.to_uncompressed()), "delta_g2": hex::encode(params.vk.delta_g2.to_uncompressed()),
                   "ic": params.vk.ic.iter().map(|p| hex::encode(p.to_uncompressed())).collect::<Vec<_>>()},
        },
        "proof": {
            "rng": "ChaCha20Rng::from_seed([9; 32]): r = Scalar::random, then s",
            "r_le": hex::encode(r.to_bytes()), "s_le": hex::encode(s.to_bytes()),
            "a": hex::encode(proof.a.to_uncompressed()), "b": hex::encode(proof.b.to_uncompressed()), "c": hex::encode(proof.c.to_uncompressed()),
        },
    })
}

// --- the aux inputs of the three test circuits, computed as the reference's test module computes them (test.rs:10-114): those helpers are
// private to `#[cfg(test)]`, so they are restated over the same public API (ZkStateBuilder + the chain's hasher)
fn update_aux(fee: Money) -> ZkScalar {
    let mut b = ZkStateBuilder::<ZkHasher>::new(ZkStateModel::Struct { field_types: vec![ZkStateModel::Scalar, ZkStateModel::Scalar] });
    b.batch_set(&ZkDeltaPairs(
        [(ZkDataLocator(vec![0]), Some(fee.token_id.into())), (ZkDataLocator(vec![1]), Some(ZkScalar::from(fee.amount)))].into(),
    ))
    .unwrap();
    b.compress().unwrap().state_hash
}
fn empty_list_root(fields: usize, log4_batch_size: u8) -> ZkScalar {
    // no enabled transition: the root of the untouched list (what deposits_root / withdraws_root return for an empty slice)
    let model = ZkStateModel::List { item_type: Box::new(ZkStateModel::Struct { field_types: vec![ZkStateModel::Scalar; fields] }), log4_size: log4_batch_size };
    ZkStateBuilder::<ZkHasher>::new(model).compress().unwrap().state_hash
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let arg = |k: &str| args.iter().position(|a| a == k).and_then(|i| args.get(i + 1)).cloned();
    if args.iter().any(|a| a == "--help") {
        eprintln!("usage: bzk-fixture-dump [--out FILE] [--params-dir DIR]");
        return;
    }
    let out = arg("--out").unwrap_or_else(|| "../../tests/golden/bellman_vectors.json".into());
    let params_dir = arg("--params-dir");
    let (c456, s123) = (ZkScalar::from(456), ZkScalar::from(123));
    let mut all = Vec::new();
    {
        let aux = update_aux(Money::ziesha(0));
        let circuit = UpdateCircuit {
            log4_tree_size: 3, log4_token_tree_size: 3, log4_update_batch_size: 1,
            commitment: c456, height: 0, state: s123, aux_data: aux, next_state: s123, fee_token: ContractId::Ziesha,
            transitions: vec![UpdateTransition::null(3, 3); 4],
        };
        all.push(dump("update_3_3_1", circuit, &[c456, ZkScalar::from(0), s123, aux, s123], params_dir.as_deref()));
    }
    {
        let aux = empty_list_root(4, 1);
        let circuit = DepositCircuit {
            log4_tree_size: 3, log4_token_tree_size: 3, log4_deposit_batch_size: 1,
            commitment: c456, height: 0, state: s123, aux_data: aux, next_state: s123,
            transitions: vec![DepositTransition::null(3, 3); 4],
        };
        all.push(dump("deposit_3_3_1", circuit, &[c456, ZkScalar::from(0), s123, aux, s123], params_dir.as_deref()));
    }
    {
        let aux = empty_list_root(7, 1);
        let circuit = WithdrawCircuit {
            log4_tree_size: 3, log4_token_tree_size: 3, log4_withdraw_batch_size: 1,
            commitment: c456, height: 0, state: s123, aux_data: aux, next_state: s123,
            transitions: vec![WithdrawTransition::null(3, 3); 4],
        };
        all.push(dump("withdraw_3_3_1", circuit, &[c456, ZkScalar::from(0), s123, aux, s123], params_dir.as_deref()));
    }
    let doc = serde_json::json!({
        "made_by": "rust/fixture-dump (bellman 0.14 / bls12_381 0.8 through ziesha-network/bazuka v0.19.20's own circuits)",
        "encodings": "scalars: Scalar::to_bytes (canonical, little-endian); points: to_uncompressed(); densities: one byte per variable",
        "circuits": all,
    });
    std::fs::write(&out, serde_json::to_string_pretty(&doc).unwrap()).expect("write json");
    eprintln!("wrote {}", out);
}
