//! bazuka-gpu: the Rust side of the drop-in boundary - Bazuka's own types over `libbzk.so` (MI355X).
//!
//! **UNVERIFIED BY COMPILATION.**  The image that builds and tests libbzk has no rustc / cargo; this file is shipped as source.
//! What IS checked mechanically (tests/test_rust_shim_cpu.py): every `sys::` symbol used here exists in `bzk-sys`, whose
//! declarations are generated from `include/bzk.h` and compared with it on every test run; argument counts of each call; the
//! reference signatures quoted in the doc comments below.  Everything else is for a maintainer with a Rust toolchain to compile.
//!
//! Each public item names the reference interface it stands beside (paths relative to ziesha-network/bazuka v0.19.20):
//!   * `GpuPoseidonHasher`      - `impl ZkHasher` (src/zk/mod.rs:152-155, 496-511) + the BULK path `hash_batch`
//!   * `groth16_prove`          - beside `groth16_verify` (src/zk/groth16/mod.rs:67-75), same argument order
//!   * `compress`               - `ZkStateModel::compress::<H>(&data)` (src/zk/mod.rs:392-399)
//!   * `DeviceStateManager`     - `KvStoreStateManager::{update_contract, root, get_data, prove}` (src/zk/state/mod.rs:218-438) for one
//!                                contract with the values resident on the GPU
//!   * `prove_work`             - an `MpnWork` (src/mpn/mod.rs:263-270) -> the `ZkProof` that `MpnWork::verify` (:281-295) accepts
//!   * `DeviceGroup`            - 1..8 GPUs of a node: window-sharded MSM + a proof pool (no torch, no Python)
//!
//! Process-level runtime settings a PROVING host should make before its first call into libbzk (they are read when the HIP runtime initialises; the
//! library cannot make them for a process that already did): `GPU_MAX_HW_QUEUES=16` (a prover keeps 4 slots x 4 streams busy) and
//! `AMD_DIRECT_DISPATCH=0` (launches through the runtime's worker threads: the prover side costs the host 0.0084 instead of 0.0202 CPU-s per proof,
//! four slots prove 4 % more per second under a CPU quota; a caller of lone MSMs keeps the default - 2 % faster for it).  `std::env::set_var` at
//! the top of `main`, as bazuka_amd/csrc/worker_main.cpp does with `setenv`; INTEGRATION.md, environment table.
//!
//! Layout assumptions (already relied upon by the reference's own `transmute`s, src/zk/groth16/mod.rs:7-17): `ZkScalar` is
//! `[u64; 4]` little-endian Montgomery limbs; bincode 1.3 with default options; `Groth16Proof` = 97 + 193 + 97 bytes under bincode.
use bazuka::core::Address;
use bazuka::mpn::MpnWork;
use bazuka::zk::groth16::Groth16Proof;
use bazuka::zk::{
    StateManagerError, ZkCompressedState, ZkDataLocator, ZkDataPairs, ZkDeltaPairs, ZkHasher, ZkLocatorError, ZkProof, ZkScalar, ZkStateModel,
};
use bzk_sys as sys;
use std::ffi::CStr;
use std::ptr;
use std::sync::{Mutex, OnceLock};

#[derive(thiserror::Error, Debug)]
pub enum GpuError {
    #[error("libbzk: {0} ({1})")]
    Status(i32, String),
    #[error("bincode: {0}")]
    Bincode(#[from] bincode::Error),
}

fn check(ctx: *mut sys::bzk_ctx, st: i32) -> Result<(), GpuError> {
    if st == sys::BZK_OK {
        return Ok(());
    }
    let detail = unsafe {
        let p = if ctx.is_null() { sys::bzk_strerror(st) } else { sys::bzk_last_error(ctx) };
        CStr::from_ptr(p).to_string_lossy().into_owned()
    };
    Err(GpuError::Status(st, detail))
}

fn scalars_ptr(v: &[ZkScalar]) -> *const u8 {
    v.as_ptr() as *const u8
}

/// One GPU + one HIP stream.  There is no CPU fallback: construction fails without a gfx950 device.
pub struct Gpu(*mut sys::bzk_ctx);
unsafe impl Send for Gpu {}

impl Gpu {
    pub fn new(device: i32) -> Result<Self, GpuError> {
        let mut h = ptr::null_mut();
        check(ptr::null_mut(), unsafe { sys::bzk_ctx_create(device, ptr::null_mut(), &mut h) })?;
        Ok(Gpu(h))
    }

    /// Bulk `ZkHasher::hash`: out[i] = poseidon(vals[i * arity .. (i + 1) * arity])  (src/zk/poseidon/mod.rs:24-84)
    pub fn hash_batch(&self, vals: &[ZkScalar], arity: usize) -> Result<Vec<ZkScalar>, GpuError> {
        assert!(arity >= 1 && arity <= 16 && vals.len() % arity == 0);
        let n = vals.len() / arity;
        let mut out = vec![ZkScalar::default(); n];
        check(self.0, unsafe { sys::bzk_poseidon_batch(self.0, scalars_ptr(vals), arity as u32, n as u64, out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }

    /// Root of a dense `ZkStateModel::List { log4_size, item_type: Scalar }` (what `ZkStateBuilder::compress` returns for it)
    pub fn merkle4_root(&self, leaves: &[ZkScalar], log4_size: u8) -> Result<ZkScalar, GpuError> {
        assert_eq!(leaves.len(), 1usize << (2 * log4_size));
        let mut root = ZkScalar::default();
        check(self.0, unsafe {
            sys::bzk_merkle4_root(self.0, scalars_ptr(leaves), log4_size as u32, &mut root as *mut _ as *mut u8, ptr::null_mut())
        })?;
        Ok(root)
    }

    /// `ZkStateModel::compress::<PoseidonHasher>(&data)` (src/zk/mod.rs:392-399) for any model over sparse pairs
    pub fn compress(&self, model: &ZkStateModel, data: &ZkDataPairs) -> Result<ZkCompressedState, DeviceStateError> {
        let m = bincode::serialize(model).map_err(GpuError::from)?;
        let d = bincode::serialize(data).map_err(GpuError::from)?;
        let mut out = [0u8; 40];
        let st = unsafe { sys::bzk_state_compress_bincode(self.0, m.as_ptr(), m.len() as u64, d.as_ptr(), d.len() as u64, out.as_mut_ptr()) };
        match st {
            sys::BZK_OK => Ok(bincode::deserialize(&out).map_err(GpuError::from)?),
            // refused exactly where the reference reports LocatorError / NonScalarLocatorError (or panics); device failures as GpuError
            e => Err(state_error(self.0, e)),
        }
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { sys::bzk_ctx_destroy(self.0) }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// `KvStoreStateManager<H>` (src/zk/state/mod.rs:29-31) for ONE contract, values resident on the device.  The reference's functions
// are static over a `KvStore` and a `ContractId`; here the handle IS the contract's store.  One `update_contract` is one batched
// plan on the GPU (every touched node hashed once) instead of depth-many hashes and KV reads per pair.
// ---------------------------------------------------------------------------------------------------------------------------
pub struct DeviceStateManager<'g> {
    st: *mut sys::bzk_state,
    gpu: &'g Gpu,
}
unsafe impl Send for DeviceStateManager<'_> {}

/// What a `DeviceStateManager` call can fail with: the reference's own `StateManagerError` where the library REFUSED the request
/// (nothing changed), or a device-side failure the reference has no variant for (out of HBM while the value store grows; a device
/// error in the middle of an update, after which the handle answers `BZK_E_DEVICE` to everything).  Never a panic.
#[derive(thiserror::Error, Debug)]
pub enum DeviceStateError {
    #[error(transparent)]
    State(#[from] StateManagerError),
    #[error(transparent)]
    Gpu(#[from] GpuError),
}

/// Maps a non-OK status of a `bzk_state_*` call.  `BZK_E_ARG` = refused: `bzk_last_refusal` says which of the reference's errors it
/// stands for (src/zk/state/mod.rs:12-27, src/zk/mod.rs:348-351); anything else is a device / allocation failure and travels as
/// `GpuError::Status` with `bzk_last_error`'s text.
fn state_error(ctx: *mut sys::bzk_ctx, st: i32) -> DeviceStateError {
    if st == sys::BZK_E_ARG {
        return DeviceStateError::State(match unsafe { sys::bzk_last_refusal(ctx) } {
            sys::BZK_REFUSE_NON_SCALAR_LOCATOR => StateManagerError::NonScalarLocatorError,
            sys::BZK_REFUSE_NON_TREE_LOCATOR => StateManagerError::NonTreeLocatorError,
            // InvalidLocator proper; a duplicate locator / a non-canonical value cannot be expressed by the Rust types that produced
            // the bincode (a HashMap key occurs once, a ZkScalar is a residue) and malformed bincode cannot come out of `serialize`:
            // all of them mean "this request names something the state does not have"
            _ => StateManagerError::LocatorError(ZkLocatorError::InvalidLocator),
        });
    }
    DeviceStateError::Gpu(check(ctx, st).unwrap_err())
}

impl<'g> DeviceStateManager<'g> {
    /// an empty contract of this model (`ZkCompressedState::empty::<H>(model)`, src/zk/mod.rs:557-562)
    pub fn new(gpu: &'g Gpu, model: &ZkStateModel) -> Result<Self, GpuError> {
        let m = bincode::serialize(model)?;
        let mut st = ptr::null_mut();
        check(gpu.0, unsafe { sys::bzk_state_create(gpu.0, m.as_ptr(), m.len() as u64, &mut st) })?;
        Ok(DeviceStateManager { st, gpu })
    }

    /// `update_contract(db, id, patch, target_height)` (src/zk/state/mod.rs:286-308): all or nothing
    pub fn update_contract(&mut self, patch: &ZkDeltaPairs, target_height: u64) -> Result<ZkCompressedState, DeviceStateError> {
        let d = bincode::serialize(patch).map_err(GpuError::from)?;
        let mut out = [0u8; 40];
        match unsafe { sys::bzk_state_update_bincode(self.st, d.as_ptr(), d.len() as u64, target_height, out.as_mut_ptr()) } {
            sys::BZK_OK => Ok(bincode::deserialize(&out).map_err(GpuError::from)?),
            e => Err(state_error(self.gpu.0, e)),
        }
    }

    /// `root` (:274-284) and `height_of` (:210-216)
    pub fn root(&self) -> Result<(ZkCompressedState, u64), DeviceStateError> {
        let (mut hash, mut size, mut height) = (ZkScalar::default(), 0u64, 0u64);
        match unsafe { sys::bzk_state_root(self.st, &mut hash as *mut _ as *mut u8, &mut size, &mut height) } {
            sys::BZK_OK => Ok((ZkCompressedState::new(hash, size), height)),
            e => Err(state_error(self.gpu.0, e)),  // BZK_E_DEVICE: an earlier update failed on the device
        }
    }

    /// `get_data` (:422-438) for many locators in one device read
    pub fn get_data(&self, locators: &[ZkDataLocator]) -> Result<Vec<ZkScalar>, DeviceStateError> {
        let mut off = vec![0u64];
        let mut flat = Vec::new();
        for l in locators {
            flat.extend_from_slice(&l.0);
            off.push(flat.len() as u64);
        }
        let mut out = vec![ZkScalar::default(); locators.len()];
        match unsafe { sys::bzk_state_get(self.st, off.as_ptr(), flat.as_ptr(), locators.len() as u64, out.as_mut_ptr() as *mut u8) } {
            sys::BZK_OK => Ok(out),
            e => Err(state_error(self.gpu.0, e)),
        }
    }

    /// `prove(db, id, tree_loc, index)` (:218-264): `Vec<[ZkScalar; 3]>`, leaf level first
    pub fn prove(&self, tree_loc: &ZkDataLocator, index: u64) -> Result<Vec<[ZkScalar; 3]>, DeviceStateError> {
        let mut log4 = 0u32;
        let st = unsafe { sys::bzk_state_prove(self.st, tree_loc.0.as_ptr(), tree_loc.0.len() as u64, ptr::null(), 0, ptr::null_mut(), &mut log4) };
        if st != sys::BZK_OK {
            return Err(state_error(self.gpu.0, st));  // InvalidLocator vs NonTreeLocatorError: told apart by bzk_last_refusal
        }
        let mut out = vec![[ZkScalar::default(); 3]; log4 as usize];
        match unsafe {
            sys::bzk_state_prove(self.st, tree_loc.0.as_ptr(), tree_loc.0.len() as u64, &index, 1, out.as_mut_ptr() as *mut u8, &mut log4)
        } {
            sys::BZK_OK => Ok(out),
            e => Err(state_error(self.gpu.0, e)),  // an index beyond the list: InvalidLocator
        }
    }

    pub fn gpu(&self) -> &Gpu {
        self.gpu
    }
}

impl Drop for DeviceStateManager<'_> {
    fn drop(&mut self) {
        unsafe { sys::bzk_state_free(self.st) }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// `impl ZkHasher` (src/zk/mod.rs:152-155): a static fn, so the context is a process-wide one on device BZK_DEVICE (default 0).
// `hash` of ONE input vector runs on the host (a single Poseidon is a ~50 us job; a PCIe round trip would cost more) through the
// library's own host Poseidon - same parameters, same bytes; the GPU is for `hash_batch`.
// ---------------------------------------------------------------------------------------------------------------------------
static GPU: OnceLock<Mutex<Gpu>> = OnceLock::new();

pub fn shared_gpu() -> &'static Mutex<Gpu> {
    GPU.get_or_init(|| {
        let dev = std::env::var("BZK_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        Mutex::new(Gpu::new(dev).expect("bazuka-gpu: no usable gfx950 device (libbzk has no CPU fallback)"))
    })
}

#[derive(Debug, Clone, PartialEq, Eq, std::hash::Hash, Default)]
pub struct GpuPoseidonHasher;

impl ZkHasher for GpuPoseidonHasher {
    const MAX_ARITY: usize = 16; // poseidon::MAX_ARITY (src/zk/poseidon/params/mod.rs:25)
    fn hash(vals: &[ZkScalar]) -> ZkScalar {
        let mut out = ZkScalar::default();
        let st = unsafe { sys::bzk_host_poseidon(scalars_ptr(vals), vals.len() as u32, &mut out as *mut _ as *mut u8) };
        assert_eq!(st, sys::BZK_OK, "arity outside 1..=16 (the reference unwraps a missing parameter set here)");
        out
    }
}

impl GpuPoseidonHasher {
    /// the bulk path: n independent hashes of the same arity in one device launch
    pub fn hash_batch(vals: &[ZkScalar], arity: usize) -> Vec<ZkScalar> {
        shared_gpu().lock().unwrap().hash_batch(vals, arity).expect("bzk_poseidon_batch")
    }
}

/// `ZkStateModel::compress` through the shared context (drop-in for `model.compress::<PoseidonHasher>(&data)`)
pub fn compress(model: &ZkStateModel, data: &ZkDataPairs) -> Result<ZkCompressedState, DeviceStateError> {
    shared_gpu().lock().unwrap().compress(model, data)
}

// ---------------------------------------------------------------------------------------------------------------------------
// Groth16: proving parameters live on the device; `groth16_prove` sits beside `groth16_verify` (src/zk/groth16/mod.rs:67-75).
// ---------------------------------------------------------------------------------------------------------------------------
/// A CRS uploaded once (bellman `Parameters`): from a bellman parameter file + the circuit shape's density maps.
pub struct ProvingParams {
    gpu: Gpu,
    params: *mut sys::bzk_params,
    /// bincode(`Groth16VerifyingKey`) as the library derived it from the file
    pub vk_bincode: Vec<u8>,
}
unsafe impl Send for ProvingParams {}

impl ProvingParams {
    /// `bytes` = `bellman::groth16::Parameters::<Bls12>::write` output; densities = views 4 / 5 of the empty circuit
    pub fn from_bellman(device: i32, bytes: &[u8], n_in: u32, n_aux: u32, a_density: &[u8], b_density: &[u8]) -> Result<Self, GpuError> {
        let gpu = Gpu::new(device)?;
        let mut params = ptr::null_mut();
        let mut vk = vec![0u8; 878 + 97 * n_in as usize];
        check(gpu.0, unsafe {
            sys::bzk_params_load_bellman(gpu.0, bytes.as_ptr(), bytes.len() as u64, n_in, n_aux, a_density.as_ptr(), b_density.as_ptr(),
                                         &mut params, vk.as_mut_ptr(), vk.len() as u64)
        })?;
        Ok(ProvingParams { gpu, params, vk_bincode: vk })
    }
}

impl Drop for ProvingParams {
    fn drop(&mut self) {
        unsafe { sys::bzk_params_free(self.gpu.0, self.params) }
    }
}

/// A synthesized circuit instance (assignment z and the evaluations A.z, B.z, C.z in pinned host memory)
pub struct Witness(*mut sys::bzk_r1cs);
unsafe impl Send for Witness {}

impl Witness {
    fn view(&self, which: i32) -> (*const u8, u64) {
        let mut bytes = 0u64;
        let p = unsafe { sys::bzk_r1cs_data(self.0, which, &mut bytes) } as *const u8;
        (p, bytes)
    }
    fn assignment(&self) -> sys::bzk_assignment {
        let (z, zb) = self.view(0);
        let (az, ab) = self.view(1);
        let (bz, _) = self.view(2);
        let (cz, _) = self.view(3);
        sys::bzk_assignment { z, az, bz, cz, n_rows: ab / 32, n_vars: zb / 32 }
    }
    /// the five public inputs [commitment, height, state, aux_data, next_state] follow z[0] = 1
    pub fn satisfied(&self) -> bool {
        let mut info = [0u64; 9];
        unsafe { sys::bzk_r1cs_info(self.0, info.as_mut_ptr()) == sys::BZK_OK && info[6] == 0 }
    }
}

impl Drop for Witness {
    fn drop(&mut self) {
        unsafe { sys::bzk_r1cs_free(self.0) }
    }
}

/// The circuit instance of an `MpnWork` for `prover` (commitment = `ZkScalar::new(sha3(bincode((prover, reward))))`,
/// src/mpn/mod.rs:283-285; transitions padded with `::null` ones as `prepare_works` leaves them)
pub fn synthesize_work(work: &MpnWork, prover: &Address) -> Result<Witness, GpuError> {
    let bytes = bincode::serialize(work)?;
    let prover_pub = bincode::serialize(prover)?; // Address = 32-byte ed25519 public key
    let mut w = ptr::null_mut();
    let mut consumed = 0u64;
    check(ptr::null_mut(), unsafe { sys::bzk_mpn_work_decode(bytes.as_ptr(), bytes.len() as u64, 0, &mut w, &mut consumed) })?;
    let mut r1cs = ptr::null_mut();
    let st = unsafe { sys::bzk_mpn_work_synthesize(w, prover_pub.as_ptr(), ptr::null(), 0, 0, &mut r1cs) };
    unsafe { sys::bzk_mpn_work_free(w) };
    check(ptr::null_mut(), st)?;
    Ok(Witness(r1cs))
}

/// Beside `groth16_verify(vk, commitment, prev_height, prev_state, aux_data, next_state, proof) -> bool`
/// (src/zk/groth16/mod.rs:67-75): the prover of the same statement.  The public inputs are those the witness was synthesized
/// with (z[1..=5]); `r`, `s` are the blinding factors bellman's `create_random_proof` draws from its rng.
pub fn groth16_prove(params: &ProvingParams, witness: &Witness, r: ZkScalar, s: ZkScalar) -> Result<Groth16Proof, GpuError> {
    let asg = witness.assignment();
    let mut buf = [0u8; 387];
    check(params.gpu.0, unsafe {
        sys::bzk_groth16_prove(params.gpu.0, params.params, &asg, &r as *const _ as *const u8, &s as *const _ as *const u8, buf.as_mut_ptr())
    })?;
    Ok(bincode::deserialize(&buf)?) // private fields: through bincode, never a pointer cast
}

/// `synthesize_work` with the hash-dependent witness values of the work's transitions LEFT TO THE DEVICE (include/bzk.h: BZK_SYNTH_DEFER): the
/// Poseidon gadget's variables, the Merkle muxes and the root checks - four fifths of a transition - are not evaluated on the host;
/// `groth16_prove_witness` completes them on the GPU before proving.  All three kinds of work (update, deposit, withdraw).
pub fn synthesize_work_deferred(work: &MpnWork, prover: &Address) -> Result<Witness, GpuError> {
    let bytes = bincode::serialize(work)?;
    let prover_pub = bincode::serialize(prover)?;
    let mut w = ptr::null_mut();
    let mut consumed = 0u64;
    check(ptr::null_mut(), unsafe { sys::bzk_mpn_work_decode(bytes.as_ptr(), bytes.len() as u64, 0, &mut w, &mut consumed) })?;
    let mut r1cs = ptr::null_mut();
    let st = unsafe { sys::bzk_mpn_work_synthesize(w, prover_pub.as_ptr(), ptr::null(), 0, sys::BZK_SYNTH_DEFER as i32, &mut r1cs) };
    unsafe { sys::bzk_mpn_work_free(w) };
    check(ptr::null_mut(), st)?;
    Ok(Witness(r1cs))
}

/// `groth16_prove` over the generator's own instance handle: a deferred instance is completed on the device first.  `BZK_E_UNSAT` (a
/// deferred constraint does not hold) arrives as `GpuError::Status`, never as a panic.
pub fn groth16_prove_witness(params: &ProvingParams, witness: &Witness, r: ZkScalar, s: ZkScalar) -> Result<Groth16Proof, GpuError> {
    let mut buf = [0u8; 387];
    check(params.gpu.0, unsafe {
        sys::bzk_groth16_prove_r1cs(params.gpu.0, params.params, witness.0, &r as *const _ as *const u8, &s as *const _ as *const u8, buf.as_mut_ptr())
    })?;
    Ok(bincode::deserialize(&buf)?)
}

/// An instance's assignment resident in HBM (`bzk_r1cs_stage`): the uploads and - for a deferred instance - the device-side fill run on a context of the
/// witness PRODUCER's, so that the prover slot only copies device to device.  The `Witness` it was staged from is kept alive inside (its pinned host arrays are
/// read by the staging stream until `wait` or a prove call has returned); the handle goes back to the staging context's pool on drop.
pub struct Staged<'g> {
    h: *mut sys::bzk_staged,
    _gpu: &'g Gpu,
    _witness: Witness,
}
unsafe impl Send for Staged<'_> {}

impl Gpu {
    /// `bzk_r1cs_stage` on THIS context's stream (a producer's context, not a prover slot's); returns at once
    pub fn stage(&self, witness: Witness) -> Result<Staged<'_>, GpuError> {
        let mut h = ptr::null_mut();
        check(self.0, unsafe { sys::bzk_r1cs_stage(self.0, witness.0, &mut h) })?;
        Ok(Staged { h, _gpu: self, _witness: witness })
    }
}

impl Staged<'_> {
    /// blocks until the staging work has finished; `BZK_E_UNSAT` (a deferred constraint does not hold, or a transition's computed state differs from the
    /// builder's prediction) arrives as `GpuError::Status`, never as a panic
    pub fn wait(&self) -> Result<(), GpuError> {
        check(ptr::null_mut(), unsafe { sys::bzk_staged_wait(self.h) })
    }
    /// one complete array as the device left it (0 z, 1 A.z, 2 B.z, 3 C.z): what `ProvingAssignment` holds after `Circuit::synthesize`
    pub fn read(&self, which: i32) -> Result<Vec<ZkScalar>, GpuError> {
        let mut bytes = 0u64;
        check(ptr::null_mut(), unsafe { sys::bzk_staged_read(self.h, which, ptr::null_mut(), 0, &mut bytes) })?;
        let mut out = vec![ZkScalar::default(); (bytes / 32) as usize];
        check(ptr::null_mut(), unsafe { sys::bzk_staged_read(self.h, which, out.as_mut_ptr() as *mut u8, bytes, ptr::null_mut()) })?;
        Ok(out)
    }
}

impl Drop for Staged<'_> {
    fn drop(&mut self) {
        // the staging stream may still be reading the witness's host arrays: wait before they (and the handle) go
        unsafe {
            let _ = sys::bzk_staged_wait(self.h);
            sys::bzk_staged_free(self.h)
        }
    }
}

/// `groth16_prove` over a staged instance (any context of the same device as the one it was staged on): waits for the staging work in stream order
/// and copies device to device; same proof bytes as `groth16_prove` / `groth16_prove_witness`
pub fn groth16_prove_staged(params: &ProvingParams, staged: &Staged<'_>, r: ZkScalar, s: ZkScalar) -> Result<Groth16Proof, GpuError> {
    let mut buf = [0u8; 387];
    check(params.gpu.0, unsafe {
        sys::bzk_groth16_prove_staged(params.gpu.0, params.params, staged.h, &r as *const _ as *const u8, &s as *const _ as *const u8, buf.as_mut_ptr())
    })?;
    Ok(bincode::deserialize(&buf)?)
}

/// `GET /bincode/mpn/work` -> this -> `POST /bincode/mpn/solution` (src/client/messages.rs:368-388): the `ZkProof` that
/// `work.verify(prover, &proof)` accepts
pub fn prove_work(params: &ProvingParams, work: &MpnWork, prover: &Address, r: ZkScalar, s: ZkScalar) -> Result<ZkProof, GpuError> {
    let w = synthesize_work(work, prover)?;
    Ok(ZkProof::Groth16(Box::new(groth16_prove(params, &w, r, s)?)))
}

// ---------------------------------------------------------------------------------------------------------------------------
// Device groups (include/bzk.h row (e)): one process drives the GPUs of a node.
// ---------------------------------------------------------------------------------------------------------------------------
pub struct DeviceGroup(*mut sys::bzk_mg);
unsafe impl Send for DeviceGroup {}

pub struct GroupBases<'g> {
    group: &'g DeviceGroup,
    h: *mut sys::bzk_mg_bases,
}

impl DeviceGroup {
    /// transport: RCCL all-gather over xGMI when the devices are distinct, host memory otherwise (`BZK_MG_X_AUTO`)
    pub fn new(devices: &[i32]) -> Result<Self, GpuError> {
        let mut h = ptr::null_mut();
        check(ptr::null_mut(), unsafe { sys::bzk_mg_create(devices.as_ptr(), devices.len() as i32, sys::BZK_MG_X_AUTO, &mut h) })?;
        Ok(DeviceGroup(h))
    }
    pub fn world(&self) -> i32 {
        unsafe { sys::bzk_mg_world(self.0) }
    }
    /// what this process could contribute to a process-per-GPU group on `device`: bit 0 = usable gfx950 device, bit 1 = librccl loadable.
    /// A host collects the answers of all ranks BEFORE creating a group on the RCCL transport (`ncclCommInitRank` blocks until every
    /// rank arrives; include/bzk.h)
    pub fn probe(device: i32) -> i32 {
        unsafe { sys::bzk_mg_probe(device) }
    }
    fn fail(&self, st: i32) -> GpuError {
        GpuError::Status(st, unsafe { CStr::from_ptr(sys::bzk_mg_last_error(self.0)).to_string_lossy().into_owned() })
    }
    /// replicate a static G1 base set (raw affine `x | y`, 96 bytes per point) on every device, converted once
    pub fn load_g1_bases(&self, raw: &[u8]) -> Result<GroupBases<'_>, GpuError> {
        let mut h = ptr::null_mut();
        let st = unsafe { sys::bzk_mg_bases_g1_load(self.0, raw.as_ptr(), (raw.len() / 96) as u64, &mut h) };
        if st != sys::BZK_OK {
            return Err(self.fail(st));
        }
        Ok(GroupBases { group: self, h })
    }
    /// one MSM sharded by scalar-window range over the group; returns the packed affine point (97 bytes)
    pub fn msm_g1(&self, bases: &GroupBases<'_>, scalars: &[ZkScalar]) -> Result<[u8; 97], GpuError> {
        let mut out = [0u8; 97];
        let st = unsafe { sys::bzk_mg_msm_g1(self.0, bases.h, scalars_ptr(scalars), scalars.len() as u64, 0, out.as_mut_ptr()) };
        if st != sys::BZK_OK {
            return Err(self.fail(st));
        }
        Ok(out)
    }
}

impl Drop for GroupBases<'_> {
    fn drop(&mut self) {
        unsafe { sys::bzk_mg_bases_free(self.group.0, self.h) }
    }
}

impl Drop for DeviceGroup {
    fn drop(&mut self) {
        unsafe { sys::bzk_mg_destroy(self.0) }
    }
}
