//! The seam: `tfhe::boolean::{gen_keys, ClientKey, ServerKey, Ciphertext}` as used by `base`
//! (base/src/primitive.rs:5-6, boolean_tree.rs:647-651), implemented on `tfhe-aes-hip-sys`.
//!
//! Ciphertexts live in the engine's device pool; the host holds slot handles.  `ServerKey::not` only flips a flag
//! on the handle: the engine folds the negation into the consumer's linear combination (`tfa_gate.neg`).
//! Trivial ciphertexts never reach the engine: every gate resolves them here with tfhe-rs' rules.
//! Errors from the engine panic, as every failure in the reference does (`unwrap`/`expect`).
use std::ffi::CStr;
use std::sync::{Arc, Mutex};
use tfhe_aes_hip_sys as sys;

#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct Slot {
    pub index: u32,
    pub negated: bool,
}

#[derive(Clone, Debug)]
pub enum Ciphertext {
    Encrypted(Slot),
    Trivial(bool),
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Op {
    And,
    Nand,
    Or,
    Nor,
    Xor,
    Xnor,
}

struct Engine {
    ctx: *mut sys::tfa_ctx,
    params: sys::tfa_params,
    next_slot: u32,
    capacity: u32,
}
unsafe impl Send for Engine {}

impl Engine {
    fn check(&self, rc: i32) {
        if rc != sys::TFA_OK {
            let msg = unsafe { CStr::from_ptr(sys::tfa_last_error(self.ctx)) };
            panic!("tfhe-aes-hip error {rc}: {}", msg.to_string_lossy());
        }
    }
    fn alloc(&mut self, count: u32) -> u32 {
        let first = self.next_slot;
        self.next_slot += count;
        if self.next_slot > self.capacity {
            self.capacity = (self.next_slot * 2).max(1 << 16);
            let rc = unsafe { sys::tfa_pool_reserve(self.ctx, self.capacity) };
            self.check(rc);
        }
        first
    }
}
impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::tfa_ctx_destroy(self.ctx) };
    }
}

/// `tfhe::boolean::ServerKey`: `Clone` and shareable between threads like the original
/// (primitive.rs:21, `&ServerKey` crossing rayon workers); submissions are serialised on the context.
#[derive(Clone)]
pub struct ServerKey {
    engine: Arc<Mutex<Engine>>,
}

pub struct ClientKey {
    params: sys::tfa_params,
    lwe_sk: Vec<u32>,
    server: ServerKey,
    deterministic: bool, // a non-zero keygen seed: reproducible, NON-SECRET encryption streams (tests, benchmarks)
    next_seed: Mutex<u64>,
}

/// `tfhe::boolean::gen_keys()` (tfhe-aes/src/main.rs:64): key material from the client routines of the library,
/// bootstrapping and keyswitching keys uploaded to `hip_device`.  seed 0 = OS entropy.
pub fn gen_keys_on(hip_device: i32, seed: u64) -> (ClientKey, ServerKey) {
    unsafe {
        let mut p = std::mem::zeroed::<sys::tfa_params>();
        sys::tfa_default_params(&mut p);
        let n = p.lwe_dimension as usize;
        let big = (p.glwe_dimension * p.polynomial_size) as usize;
        let mut lwe_sk = vec![0u32; n];
        let mut glwe_sk = vec![0u32; big];
        let mut bsk = vec![0u32; sys::tfa_bsk_words(&p) as usize];
        let mut ksk = vec![0u32; sys::tfa_ksk_words(&p) as usize];
        let rc = sys::tfa_client_keygen(&p, seed, lwe_sk.as_mut_ptr(), glwe_sk.as_mut_ptr(), bsk.as_mut_ptr(), ksk.as_mut_ptr());
        assert_eq!(rc, sys::TFA_OK, "tfa_client_keygen failed (no OS entropy?)");
        let mut ctx = std::ptr::null_mut();
        let rc = sys::tfa_ctx_create(&p, hip_device, &mut ctx);
        assert_eq!(rc, sys::TFA_OK, "tfa_ctx_create failed: is a gfx950 GPU visible?");
        let engine = Engine { ctx, params: p, next_slot: 0, capacity: 0 };
        engine.check(sys::tfa_keys_upload(ctx, bsk.as_ptr(), ksk.as_ptr()));
        let server = ServerKey { engine: Arc::new(Mutex::new(engine)) };
        (ClientKey { params: p, lwe_sk, server: server.clone(), deterministic: seed != 0, next_seed: Mutex::new(seed.wrapping_mul(0x9E37_79B9_7F4A_7C15)) }, server)
    }
}
pub fn gen_keys() -> (ClientKey, ServerKey) {
    gen_keys_on(0, 0)
}

impl ClientKey {
    /// `ClientKey::encrypt` (primitive.rs:63,71): fresh LWE encryption of +-1/8, written straight into the pool.
    pub fn encrypt(&self, bit: bool) -> Ciphertext {
        self.encrypt_many(&[bit]).pop().unwrap()
    }
    pub fn encrypt_many(&self, bits: &[bool]) -> Vec<Ciphertext> {
        let n1 = self.params.lwe_dimension as usize + 1;
        let raw: Vec<u8> = bits.iter().map(|&b| b as u8).collect();
        let mut lwe = vec![0u32; bits.len() * n1];
        let seed = {
            let mut s = self.next_seed.lock().unwrap();
            *s = s.wrapping_add(1);
            if self.deterministic { (*s).max(1) } else { 0 }
        };
        let rc = unsafe { sys::tfa_client_encrypt(&self.params, self.lwe_sk.as_ptr(), seed, raw.as_ptr(), bits.len() as u32, lwe.as_mut_ptr()) };
        assert_eq!(rc, sys::TFA_OK);
        let mut e = self.server.engine.lock().unwrap();
        let first = e.alloc(bits.len() as u32);
        let slots: Vec<u32> = (first..first + bits.len() as u32).collect();
        let rc = unsafe { sys::tfa_pool_write(e.ctx, slots.as_ptr(), slots.len() as u32, lwe.as_ptr()) };
        e.check(rc);
        slots.into_iter().map(|index| Ciphertext::Encrypted(Slot { index, negated: false })).collect()
    }
    /// `ClientKey::decrypt` (primitive.rs:85): phase < 2^31.
    pub fn decrypt(&self, ct: &Ciphertext) -> bool {
        self.decrypt_many(std::slice::from_ref(ct))[0]
    }
    pub fn decrypt_many(&self, cts: &[Ciphertext]) -> Vec<bool> {
        let n1 = self.params.lwe_dimension as usize + 1;
        let enc: Vec<(usize, Slot)> = cts.iter().enumerate().filter_map(|(i, c)| if let Ciphertext::Encrypted(s) = c { Some((i, *s)) } else { None }).collect();
        let mut out: Vec<bool> = cts.iter().map(|c| matches!(c, Ciphertext::Trivial(true))).collect();
        if enc.is_empty() {
            return out;
        }
        let slots: Vec<u32> = enc.iter().map(|(_, s)| s.index).collect();
        let mut lwe = vec![0u32; slots.len() * n1];
        {
            let e = self.server.engine.lock().unwrap();
            let rc = unsafe { sys::tfa_pool_read(e.ctx, slots.as_ptr(), slots.len() as u32, lwe.as_mut_ptr()) };
            e.check(rc);
        }
        let mut bits = vec![0u8; slots.len()];
        let rc = unsafe { sys::tfa_client_decrypt(&self.params, self.lwe_sk.as_ptr(), lwe.as_ptr(), slots.len() as u32, bits.as_mut_ptr()) };
        assert_eq!(rc, sys::TFA_OK);
        for ((i, s), b) in enc.iter().zip(bits) {
            out[*i] = (b != 0) != s.negated; // a negated handle decrypts to the complement
        }
        out
    }
}

/// Index of a gate's result in `StageBatch::run`'s output.
pub type Ticket = usize;

enum Pending {
    Ready(Ciphertext), // resolved on the host (trivial operands)
    Gate,              // result in the slot reserved for this ticket
}

/// The gates of ONE stage (no gate may read another's output).  Replaces a rayon `par_iter` over gate calls:
/// record with `gate`/`mux`, then `run` submits them with a single `tfa_stage_submit`.
pub struct StageBatch<'a> {
    key: &'a ServerKey,
    pending: Vec<Pending>,
    gates: Vec<sys::tfa_gate>, // `out` holds the ticket until `run` assigns slots
}

impl ServerKey {
    pub fn trivial_encrypt(&self, bit: bool) -> Ciphertext {
        Ciphertext::Trivial(bit)
    }
    pub fn not(&self, ct: &Ciphertext) -> Ciphertext {
        match ct {
            Ciphertext::Trivial(b) => Ciphertext::Trivial(!b),
            Ciphertext::Encrypted(s) => Ciphertext::Encrypted(Slot { index: s.index, negated: !s.negated }),
        }
    }
    pub fn not_assign(&self, ct: &mut Ciphertext) {
        *ct = self.not(ct);
    }
    pub fn batch(&self) -> StageBatch<'_> {
        StageBatch { key: self, pending: Vec::new(), gates: Vec::new() }
    }
    fn single(&self, op: Op, a: &Ciphertext, b: &Ciphertext) -> Ciphertext {
        let mut batch = self.batch();
        let t = batch.gate(op, a, b);
        batch.run().swap_remove(t)
    }
    pub fn and(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::And, a, b) }
    pub fn nand(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::Nand, a, b) }
    pub fn or(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::Or, a, b) }
    pub fn nor(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::Nor, a, b) }
    pub fn xor(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::Xor, a, b) }
    pub fn xnor(&self, a: &Ciphertext, b: &Ciphertext) -> Ciphertext { self.single(Op::Xnor, a, b) }
    pub fn and_assign(&self, a: &mut Ciphertext, b: &Ciphertext) { *a = self.and(a, b); }
    pub fn or_assign(&self, a: &mut Ciphertext, b: &Ciphertext) { *a = self.or(a, b); }
    pub fn xor_assign(&self, a: &mut Ciphertext, b: &Ciphertext) { *a = self.xor(a, b); }
    pub fn mux(&self, c: &Ciphertext, t: &Ciphertext, e: &Ciphertext) -> Ciphertext {
        let mut batch = self.batch();
        let k = batch.mux(c, t, e);
        batch.run().swap_remove(k)
    }
}

fn clear(op: Op, a: bool, b: bool) -> bool {
    match op {
        Op::And => a & b,
        Op::Nand => !(a & b),
        Op::Or => a | b,
        Op::Nor => !(a | b),
        Op::Xor => a ^ b,
        Op::Xnor => !(a ^ b),
    }
}

impl<'a> StageBatch<'a> {
    fn ready(&mut self, ct: Ciphertext) -> Ticket {
        self.pending.push(Pending::Ready(ct));
        self.pending.len() - 1
    }
    /// One two-input gate.  A trivial operand is resolved as tfhe-rs does: and(x,T)=x, and(x,F)=F, or(x,T)=T,
    /// or(x,F)=x, xor(x,T)=not x, xor(x,F)=x, and the negated forms accordingly.
    pub fn gate(&mut self, op: Op, a: &Ciphertext, b: &Ciphertext) -> Ticket {
        use Ciphertext::*;
        match (a, b) {
            (Trivial(x), Trivial(y)) => self.ready(Trivial(clear(op, *x, *y))),
            (Encrypted(_), Trivial(y)) | (Trivial(y), Encrypted(_)) => {
                let x = if let Encrypted(_) = a { a } else { b };
                // the gate as a function of x alone: constant, x, or not x
                let (f0, f1) = (clear(op, false, *y), clear(op, true, *y));
                let r = if f0 == f1 { Trivial(f0) } else if f1 { x.clone() } else { self.key.not(x) };
                self.ready(r)
            }
            (Encrypted(x), Encrypted(y)) => {
                let code = match op {
                    Op::And => sys::TFA_AND,
                    Op::Nand => sys::TFA_NAND,
                    Op::Or => sys::TFA_OR,
                    Op::Nor => sys::TFA_NOR,
                    Op::Xor => sys::TFA_XOR,
                    Op::Xnor => sys::TFA_XNOR,
                };
                self.pending.push(Pending::Gate);
                let ticket = self.pending.len() - 1;
                self.gates.push(sys::tfa_gate { op: code, in0: x.index, in1: y.index, in2: 0, out: ticket as u32, neg: x.negated as u32 | (y.negated as u32) << 1 });
                ticket
            }
        }
    }
    /// `ServerKey::mux(cond, then, else)`; trivial operands as in tfhe-rs: a trivial condition selects,
    /// mux(c,T,e)=or(c,e), mux(c,F,e)=and(not c,e), mux(c,t,T)=or(t,not c), mux(c,t,F)=and(c,t).
    pub fn mux(&mut self, c: &Ciphertext, t: &Ciphertext, e: &Ciphertext) -> Ticket {
        use Ciphertext::*;
        match (c, t, e) {
            (Trivial(s), _, _) => self.ready(if *s { t.clone() } else { e.clone() }),
            (_, Trivial(true), _) => self.gate(Op::Or, c, e),
            (_, Trivial(false), _) => { let nc = self.key.not(c); self.gate(Op::And, &nc, e) }
            (_, _, Trivial(true)) => { let nc = self.key.not(c); self.gate(Op::Or, t, &nc) }
            (_, _, Trivial(false)) => self.gate(Op::And, c, t),
            (Encrypted(s), Encrypted(x), Encrypted(y)) => {
                self.pending.push(Pending::Gate);
                let ticket = self.pending.len() - 1;
                self.gates.push(sys::tfa_gate {
                    op: sys::TFA_MUX, in0: s.index, in1: x.index, in2: y.index, out: ticket as u32,
                    neg: s.negated as u32 | (x.negated as u32) << 1 | (y.negated as u32) << 2,
                });
                ticket
            }
        }
    }
    /// Submits the recorded gates as one stage; result k belongs to ticket k.
    pub fn run(mut self) -> Vec<Ciphertext> {
        let mut e = self.key.engine.lock().unwrap();
        let first = e.alloc(self.gates.len() as u32);
        let mut slot_of = vec![0u32; self.pending.len()];
        for (k, g) in self.gates.iter_mut().enumerate() {
            slot_of[g.out as usize] = first + k as u32;
            g.out = first + k as u32;
        }
        if !self.gates.is_empty() {
            let rc = unsafe { sys::tfa_stage_submit(e.ctx, self.gates.as_ptr(), self.gates.len() as u32) };
            e.check(rc);
        }
        self.pending
            .into_iter()
            .enumerate()
            .map(|(k, p)| match p {
                Pending::Ready(ct) => ct,
                Pending::Gate => Ciphertext::Encrypted(Slot { index: slot_of[k], negated: false }),
            })
            .collect()
    }
}
