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

/// tests/r1cs.rs:620-652 (the gadget lives in the reference's test file, not in the crate: restated here against its API)
fn range_proof<CS: ConstraintSystem>(cs: &mut CS, mut v: LinearCombination, v_assignment: Option<u64>, n: usize) -> Result<(), R1CSError> {
    let mut exp_2 = Scalar::one();
    for i in 0..n {
        let (a, b, o) = cs.allocate_multiplier(v_assignment.map(|q| {
            let bit: u64 = (q >> i) & 1;
            ((1 - bit).into(), bit.into())
        }))?;
        cs.constrain(o.into());
        cs.constrain(a + (b - Scalar::one()));
        v = v - b * exp_2;
        exp_2 = exp_2 + exp_2;
    }
    cs.constrain(v);
    Ok(())
}

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
    // 4. InnerProductProof::create at n = 4 and 64 (the reference's own test shape, inner_product_proof.rs:507-583, with
    //    deterministic operands): L, R as affine bytes AND as StarkPoint::to_bytes, the final a, b, the proof's wire bytes, the
    //    challenges u_j^2, u_j^-2 and the s vector of verification_scalars, and P, so that InnerProductProof::verify can be replayed
    let pc = PedersenGens::default();
    let ipp: Vec<_> = [4usize, 64].iter().map(|&n| {
        let bp = BulletproofGens::new(n, 1);
        let share = bp.share(0);
        let g_vec: Vec<StarkPoint> = share.G(n).cloned().collect();
        let h_vec: Vec<StarkPoint> = share.H(n).cloned().collect();
        let q = det_scalar(999) * g;
        let a: Vec<Scalar> = (0..n as u64).map(|i| det_scalar(100 + i)).collect();
        let b: Vec<Scalar> = (0..n as u64).map(|i| det_scalar(300 + i)).collect();
        let g_factors: Vec<Scalar> = vec![Scalar::one(); n];
        let y_inv = det_scalar(555);
        let h_factors: Vec<Scalar> = std::iter::successors(Some(Scalar::one()), |p| Some(*p * y_inv)).take(n).collect();   // util::exp_iter (crate-private)
        let c = mpc_bulletproof::inner_product(&a, &b);
        let b_prime: Vec<Scalar> = b.iter().zip(h_factors.iter()).map(|(bi, yi)| *bi * *yi).collect();
        let p = StarkPoint::msm(&[a.clone(), b_prime, vec![c]].concat(), &[g_vec.clone(), h_vec.clone(), vec![q]].concat());
        let mut tp = Transcript::new(b"innerproducttest");
        let proof = InnerProductProof::create(&mut tp, &q, &g_factors, &h_factors, g_vec.clone(), h_vec.clone(), a.clone(), b.clone());
        let mut tv = Transcript::new(b"innerproducttest");
        let (u_sq, u_inv_sq, s) = proof.verification_scalars(n, &mut tv).expect("verification_scalars");
        let mut tv2 = Transcript::new(b"innerproducttest");
        assert!(proof.verify(n, &mut tv2, g_factors.iter().cloned(), h_factors.iter().cloned(), &p, &q, &g_vec, &h_vec).is_ok());
        json!({"n": n, "label": hx(b"innerproducttest"), "Q": hx(&point_xy(&q)), "P": hx(&point_xy(&p)),
               "a": a.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>(), "b": b.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>(),
               "y_inv": hx(&scalar_le(&y_inv)),
               "L": proof.L_vec.iter().map(|x| hx(&point_xy(x))).collect::<Vec<_>>(), "R": proof.R_vec.iter().map(|x| hx(&point_xy(x))).collect::<Vec<_>>(),
               "L_wire": proof.L_vec.iter().map(|x| hx(&x.to_bytes())).collect::<Vec<_>>(),
               "a_out": hx(&scalar_le(&proof.a)), "b_out": hx(&scalar_le(&proof.b)), "proof_wire": hx(&proof.to_bytes()),
               "u_sq": u_sq.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>(), "u_inv_sq": u_inv_sq.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>(),
               "s": s.iter().map(|x| hx(&scalar_le(x))).collect::<Vec<_>>()})
    }).collect();
    std::fs::write(format!("{out}/ref_ipp.json"), serde_json::to_string_pretty(&json!({"create": ipp})).unwrap()).unwrap();
    // 5. R1CS: the range gadget (tests/r1cs.rs:620-703) at n = 8 and 64 -- commitment, proof (wire bytes and element by element),
    //    the verifier's verdict; the prover's blinding factors come from thread_rng, so every run writes a different, equally valid
    //    instance: what the fixture pins is that OUR verifier accepts a proof the REFERENCE made (and rejects its tampered copy).
    //    The challenges and the 13 + m + 2n + 2k mega_check scalars are internal to Verifier::verify (verifier.rs:432-455, 517-532):
    //    INTEGRATION.md section 7 shows the two-line instrumentation that prints them; with it, paste them under "challenges" /
    //    "mega_check_scalars" of each record and the test compares those too.
    let r1cs: Vec<_> = [(8usize, 201u64), (64, 0x0123_4567_89ab_cdefu64)].iter().map(|&(n, v)| {
        let bp = BulletproofGens::new(n, 1);
        let mut pt = Transcript::new(b"RangeProofTest");
        let mut prover = Prover::new(&pc, &mut pt);
        let (com, var) = prover.commit(Scalar::from(v), det_scalar(4242 + n as u64));
        range_proof(&mut prover, var.into(), Some(v), n).expect("gadget");
        let proof = prover.prove(&bp).expect("prove");
        let mut vt = Transcript::new(b"RangeProofTest");
        let mut verifier = Verifier::new(&pc, &mut vt);
        let vvar = verifier.commit(com);
        range_proof(&mut verifier, vvar.into(), None, n).expect("gadget");
        let ok = verifier.verify(&proof, &bp).is_ok();
        let pts = |v: &[StarkPoint]| v.iter().map(|x| hx(&point_xy(x))).collect::<Vec<_>>();
        json!({"n_bits": n, "v": v, "label": hx(b"RangeProofTest"), "commitment": hx(&point_xy(&com)), "commitment_wire": hx(&com.to_bytes()),
               "proof_wire": hx(&proof.to_bytes()), "ok": ok,
               "points": pts(&[proof.A_I1, proof.A_O1, proof.S1, proof.A_I2, proof.A_O2, proof.S2, proof.T_1, proof.T_3, proof.T_4, proof.T_5, proof.T_6]),
               "t_x": hx(&scalar_le(&proof.t_x)), "t_x_blinding": hx(&scalar_le(&proof.t_x_blinding)), "e_blinding": hx(&scalar_le(&proof.e_blinding)),
               "ipp_L": pts(&proof.ipp_proof.L_vec), "ipp_R": pts(&proof.ipp_proof.R_vec),
               "ipp_a": hx(&scalar_le(&proof.ipp_proof.a)), "ipp_b": hx(&scalar_le(&proof.ipp_proof.b))})
    }).collect();
    std::fs::write(format!("{out}/ref_r1cs.json"), serde_json::to_string_pretty(&json!({"range": r1cs})).unwrap()).unwrap();
    eprintln!("wrote {out}/ref_curve.json, ref_msm.json, ref_transcript.json, ref_ipp.json, ref_r1cs.json");
}
