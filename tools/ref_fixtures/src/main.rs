//! Prints known-answer vectors of threshold_crypto 0.4.0 for the implementation-defined parts of the
//! hot path (the H-spec of SURVEY.md 8c): hash_g2, SecretKey::sign, encrypt_with_rng (whose `v` is
//! xor_with_hash's keystream) and the deterministic key draw.  One record per line:
//!     <kind> <field>=<hex> ...
//! All points are in the 48/96-byte COMPRESSED Zcash form (the only one the crate's public API exposes:
//! to_bytes, src/lib.rs:149-153, 255-259; hash_g2 through group::CurveAffine::into_compressed).
use group::{CurveAffine, CurveProjective};
use rand::{Rng, SeedableRng};
use rand_chacha::ChaChaRng;
use threshold_crypto::{hash_g2, Ciphertext, SecretKey};

fn seed(tag: u8) -> [u8; 32] {
    let mut s = [0u8; 32];
    for (i, b) in s.iter_mut().enumerate() {
        *b = tag.wrapping_mul(31).wrapping_add(i as u8);
    }
    s
}

fn main() {
    // hash_g2 (src/lib.rs:691-694) on the messages SURVEY 8c lists
    let mut msgs: Vec<Vec<u8>> = vec![b"".to_vec(), b"a".to_vec(), b"Test message".to_vec(), (0u8..=255).collect()];
    for j in 0u64..4 {
        let mut m = b"tc/msg".to_vec(); // the bench workload's messages
        m.extend_from_slice(&j.to_le_bytes());
        msgs.push(m);
    }
    for m in &msgs {
        let h = hash_g2(m).into_affine().into_compressed();
        println!("hash_g2 msg={} out={}", hex::encode(m), hex::encode(h.as_ref()));
    }
    // SecretKey drawn from a fixed ChaChaRng seed (Distribution<SecretKey>, src/lib.rs:324-333: Fr::random),
    // its public key and signatures (sign, src/lib.rs:379-381)
    for tag in 1u8..=3 {
        let mut rng = ChaChaRng::from_seed(seed(tag));
        let sk: SecretKey = rng.gen();
        let pk = sk.public_key();
        // reveal() prints "SecretKey(Fr(0x<64 hex digits, big-endian>))" (src/lib.rs:396-398)
        let rev = sk.reveal();
        let fr_be = rev.trim_start_matches("SecretKey(Fr(0x").trim_end_matches("))").to_string();
        println!("key seed={} sk_be={} pk={}", hex::encode(seed(tag)), fr_be, hex::encode(&pk.to_bytes()[..]));
        for m in msgs.iter().take(4) {
            let sig = sk.sign(m);
            assert!(pk.verify(&sig, m));
            println!("sign sk_be={} msg={} sig={}", fr_be, hex::encode(m), hex::encode(&sig.to_bytes()[..]));
        }
        // encrypt_with_rng (src/lib.rs:128-137) continues the SAME rng stream: r = Fr::random(rng).
        // bincode of Ciphertext(G1, Vec<u8>, G2) = 48 bytes || u64 LE length || v || 96 bytes
        // (src/serde_impl.rs:174-185 writes the compressed bytes as a fixed-size tuple).
        let msg = b"Muffins in the canteen today! Don't tell Bob.";
        let ct: Ciphertext = pk.encrypt_with_rng(&mut rng, &msg[..]);
        assert!(ct.verify());
        assert_eq!(sk.decrypt(&ct).as_deref(), Some(&msg[..]));
        let enc = bincode::serialize(&ct).expect("bincode");
        println!("encrypt seed={} sk_be={} msg={} ciphertext_bincode={}", hex::encode(seed(tag)), fr_be, hex::encode(&msg[..]), hex::encode(&enc));
    }
    // keep the linker honest about the projective import on toolchains that warn
    let _ = <pairing::bls12_381::G2 as CurveProjective>::zero();
}
