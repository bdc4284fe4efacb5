//! Reference-pinned golden vectors: run THIS against the real crate (mpc-stark 0.2 + the merlin fork) and commit what it
//! writes to tests/golden/ref_*.json -- the only route by which the oracle's parity can move from "partial" (pinned by
//! the reference's scalar-field KATs and one weight matrix) to "green": EC points, MSM outputs, point wire bytes and
//! transcript challenges of the real implementation.  tests/test_oracle.py::test_reference_fixtures picks the files up when
//! they exist.  UNCOMPILED in the build image (no Rust toolchain, crates not vendored).
//!
//!   cargo run --release --example gen_fixtures -- ../tests/golden
use mpc_bulletproof::{r1cs::*, BulletproofGens, InnerProductProof, PedersenGens};
use mpc_stark::algebra::{scalar::Scalar, stark_curve::StarkPoint};
use bpgpu_shim::{point_xy, scalar_le};
use merlin::HashChainTranscript as Transcript;
use serde_json::json;

fn hx(b: &[u8]) -> String { hex::encode(b) }
fn det_scalar(i: u64) -> Scalar { Scalar::from(0x9E3779B97F4A7C15u64.wrapping_mul(i + 1)) * Scalar::from(i + 7).inverse() }

fn main() {
    let out = std::env::args().nth(1).unwrap_or_else(|| "../tests/golden".into());
    // 1. curve arithmetic: generator multiples, sums, the 32-byte wire encoding (StarkPoint::to_bytes)
    let g = StarkPoint::generator();
    let pts: Vec<StarkPoint> = (0..8u64).map(|i| det_scalar(i) * g).collect();
    let curve = json!({
        "generator": hx(&point_xy(&g)),
        "multiples": (0..8u64).map(|i| json!({"k": hx(&scalar_le(&det_scalar(i))), "xy": hx(&point_xy(&pts[i as usize])),
                                              "wire": hx(&pts[i as usize].to_bytes())})).collect::<Vec<_>>(),
        "sum_0_1": hx(&point_xy(&(pts[0] + pts[1]))), "double_2": hx(&point_xy(&(pts[2] + pts[2]))),
        "identity_wire": hx(&StarkPoint::identity().to_bytes()),
    });
    std::fs::write(format!("{out}/ref_curve.json"), serde_json::to_string_pretty(&curve).unwrap()).unwrap();
    // 2. MSM: StarkPoint::msm at 1, 7, 154 and 1000 terms over the crate's own generators
    let bp = BulletproofGens::new(512, 1);
    let share = bp.share(0);
    let gens: Vec<StarkPoint> = share.G(512).cloned().chain(share.H(512).cloned()).collect();
    let msm: Vec<_> = [1usize, 7, 154, 1000].iter().map(|&n| {
        let s: Vec<Scalar> = (0..n as u64).map(det_scalar).collect();
        let r = StarkPoint::msm(&s, &gens[..n]);
        json!({"n": n, "scalars": s.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>(), "out": hx(&point_xy(&r))})
    }).collect();
    std::fs::write(format!("{out}/ref_msm.json"), serde_json::to_string_pretty(&json!({
        "G": gens[..512].iter().map(|p| hx(&point_xy(p))).collect::<Vec<_>>(),
        "H": gens[512..].iter().map(|p| hx(&point_xy(p))).collect::<Vec<_>>(), "cases": msm})).unwrap()).unwrap();
    // 3. transcript: challenges of a fixed script (labels / order of transcript.rs), which pins HashChainTranscript
    let mut t = Transcript::new(b"RangeProofTest");
    t.append_message(b"dom-sep", b"r1cs v1");
    t.append_u64(b"m", 1);
    let mut c = [0u8; 64];
    t.challenge_bytes(b"y", &mut c);
    std::fs::write(format!("{out}/ref_transcript.json"), serde_json::to_string_pretty(&json!({"challenge_y_wide": hx(&c)})).unwrap()).unwrap();
    // 4. InnerProductProof::create / verification_scalars at n = 4 and 64, and an R1CS range proof (n = 8, 64) with the
    //    proof bytes, commitments, every challenge and the 13 + m + 2n + 2k mega_check scalars: emitted by instrumenting
    //    verifier.rs:516 (print `scalars` there) -- see INTEGRATION.md section 7.
    let pc = PedersenGens::default();
    let mut pt = Transcript::new(b"RangeProofTest");
    let mut prover = Prover::new(&pc, &mut pt);
    let (com, var) = prover.commit(Scalar::from(201u64), Scalar::from(77u64));
    let _ = (com, var, InnerProductProof::serialized_size);
    eprintln!("wrote {out}/ref_curve.json, ref_msm.json, ref_transcript.json");
}
