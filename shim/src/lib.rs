//! Thin safe layer over `sys` (generated from include/bpgpu.h): a context handle, the boundary encodings of
//! include/bpgpu.h for `mpc_stark::algebra::{Scalar, StarkPoint}`, and the calls that replace the reference's
//! `StarkPoint::msm_iter` / `Scalar::batch_inverse` call sites.
//! UNCOMPILED in the build image (no Rust toolchain); the same ABI is exercised by the C++ mirror and by ctypes.
pub mod sys;

use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::{BigInteger, PrimeField};
use mpc_stark::algebra::{scalar::Scalar, stark_curve::StarkPoint};
use std::os::raw::c_int;

#[derive(Debug)]
pub enum Error { Arg, Len, Device, Oom, Gens, Other(c_int) }
fn ck(rc: c_int) -> Result<(), Error> {
    match rc {
        sys::BPGPU_OK => Ok(()),
        sys::BPGPU_E_ARG => Err(Error::Arg),      // malformed input: ProofError::FormatError / VerificationError at the call site
        sys::BPGPU_E_LEN => Err(Error::Len),      // the reference's assert!/panic! (inner_product_proof.rs:62-70)
        sys::BPGPU_E_DEVICE => Err(Error::Device),
        sys::BPGPU_E_OOM => Err(Error::Oom),
        sys::BPGPU_E_GENS => Err(Error::Gens),    // R1CSError::InvalidGeneratorsLength
        x => Err(Error::Other(x)),
    }
}

/// One per rayon worker / fabric executor thread (a shared one is safe but serialises).
pub struct Ctx(*mut sys::bpgpu_ctx);
unsafe impl Send for Ctx {}
impl Ctx {
    pub fn new(device: i32) -> Result<Self, Error> {
        let mut p = std::ptr::null_mut();
        ck(unsafe { sys::bpgpu_create(device, &mut p) })?;
        Ok(Ctx(p))
    }
    pub fn raw(&self) -> *mut sys::bpgpu_ctx { self.0 }
}
impl Drop for Ctx { fn drop(&mut self) { unsafe { sys::bpgpu_destroy(self.0) } } }

/// scalar = 32-byte little-endian canonical integer (what transcript.rs:87-92 absorbs)
pub fn scalar_le(s: &Scalar) -> [u8; 32] {
    let mut b = s.to_bytes_be();
    b.reverse();
    b.try_into().expect("32 bytes")
}
pub fn scalar_from_le(b: &[u8; 32]) -> Scalar {
    let mut be = *b;
    be.reverse();
    Scalar::from_be_bytes_mod_order(&be)
}
/// point = affine x || y, 32-byte little-endian each, 64 zero bytes = identity (util.rs:274-289)
pub fn point_xy(p: &StarkPoint) -> [u8; 64] {
    let mut out = [0u8; 64];
    let a = p.to_affine();                       // mpc-stark: StarkPoint wraps ark_ec Projective<StarkCurveConfig>
    if let Some((x, y)) = a.xy() {
        out[..32].copy_from_slice(&x.into_bigint().to_bytes_le());
        out[32..].copy_from_slice(&y.into_bigint().to_bytes_le());
    }
    out
}

/// StarkPoint::msm_iter(scalars, points)  (verifier.rs:516, prover.rs:465-564, inner_product_proof.rs:90-172)
pub fn msm(ctx: &Ctx, scalars: &[Scalar], points: &[StarkPoint]) -> Result<[u8; 64], Error> {
    assert_eq!(scalars.len(), points.len());
    let s: Vec<u8> = scalars.iter().flat_map(scalar_le).collect();
    let p: Vec<u8> = points.iter().flat_map(point_xy).collect();
    let mut out = [0u8; 64];
    ck(unsafe { sys::bpgpu_msm(ctx.0, s.as_ptr(), p.as_ptr(), points.len(), out.as_mut_ptr()) })?;
    Ok(out)
}
/// Scalar::batch_inverse(&mut [Scalar])  (inner_product_proof.rs:283)
pub fn batch_inverse(ctx: &Ctx, scalars: &mut [Scalar]) -> Result<(), Error> {
    let mut s: Vec<u8> = scalars.iter().flat_map(scalar_le).collect();
    ck(unsafe { sys::bpgpu_batch_inverse(ctx.0, s.as_mut_ptr(), scalars.len()) })?;
    for (i, x) in scalars.iter_mut().enumerate() {
        *x = scalar_from_le(s[32 * i..32 * i + 32].try_into().unwrap());
    }
    Ok(())
}
