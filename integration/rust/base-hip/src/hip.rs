//! The seam: `tfhe::boolean::{gen_keys, ClientKey, ServerKey, Ciphertext}` as used by `base`
//! (base/src/primitive.rs:5-6, boolean_tree.rs:647-651), implemented on `tfhe-aes-hip-sys`.
//!
//! Ciphertexts live in the engine's device pool; the host holds reference-counted slot handles, and a slot goes back
//! to the pool's free list when its last handle is dropped.  `ServerKey::not` only flips a flag on the handle: the
//! engine folds the negation into the consumer's linear combination (`tfa_gate.neg`).  Trivial ciphertexts never
//! reach the engine: every gate resolves them here with tfhe-rs' rules.  Errors from the engine panic, as every
//! failure in the reference does (`unwrap`/`expect`).
//!
//! Two ways of running gates, chosen per `ServerKey` (`set_recording`):
//! * recording (the default): a gate call only records the gate at its dependency level, 1 + the deepest level of
//!   its encrypted operands; `flush` (implied by every decrypt) submits level by level, one `tfa_stage_submit`
//!   each.  Every gate that is independent at a tree height, across the 16 bytes of a state AND across all blocks
//!   recorded since the last flush, lands in the same stage, whichever thread recorded it: the rayon loop over
//!   blocks of modes/src/ctr.rs:26-35 and the serial loop of tfhe-aes/src/main.rs:155-157 both end up sharing
//!   stages between blocks.
//! * eager: `StageBatch::run` submits its gates at once as one stage (what a caller that interleaves decryptions
//!   with gates, or wants one stage per call, asks for).
use std::ffi::CStr;
use std::sync::atomic::{AtomicU64, Ordering};
use std::sync::{Arc, Mutex};
use tfhe_aes_hip_sys as sys;

/// Pending gates beyond which a recording is run before the next gate is taken: about 28 AES blocks in flight,
/// 6.5 GB of pool.  Results do not change; gates on either side of the cut stop sharing stages.
pub const AUTO_FLUSH_GATES: usize = 2_000_000;

/// One pool slot, owned by the handles that name it.  `level` / `epoch`: the stage of the current recording that
/// writes the slot (0 once the recording that wrote it has been flushed, i.e. when `epoch` is not the key's).
pub struct SlotRef {
    pub index: u32,
    level: u32,
    epoch: u64,
    released: Arc<Mutex<Vec<u32>>>,
}
impl Drop for SlotRef {
    fn drop(&mut self) {
        // only the free list's own lock is taken here, never the engine's: a handle may be dropped anywhere
        if let Ok(mut r) = self.released.lock() {
            r.push(self.index);
        }
    }
}
impl std::fmt::Debug for SlotRef {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "slot {} (level {}, epoch {})", self.index, self.level, self.epoch)
    }
}

#[derive(Clone, Debug)]
pub struct Slot {
    pub handle: Arc<SlotRef>,
    pub negated: bool,
}
impl Slot {
    pub fn index(&self) -> u32 {
        self.handle.index
    }
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

/// What the host side did since the key was made (the counters the C++ mirror reports as `tfh_host_stats`).
#[derive(Clone, Copy, Debug, Default)]
pub struct HostStats {
    pub gates: u64,
    pub stages: u64,
    pub flushes: u64,
    pub max_stage_width: u64,
}

/// Re-associates chains of xor / xnor gates whose intermediate results nobody can observe: an intermediate whose last
/// handle is gone (its slot is in `dead`) and that is read by exactly one recorded gate, itself an xor / xnor.  Such a chain
/// is one n-input parity: it is rebuilt with the same n - 1 xor gates, on the same slots, as the tree of least depth for the
/// levels at which its operands become available (the two earliest are combined first), and every gate of the recording is
/// levelled again.  MixColumns followed by AddRoundKey (state.rs:72-118, 217-224), five dependent xor layers per output
/// bit, becomes three levels: one ECB block takes 99 stages instead of 117.  Operand negations and xnors are gathered into
/// one negation at the root.  Gate counts, gate kinds and decrypted results do not change.
fn rebalance_xor_chains(levels: Vec<Vec<sys::tfa_gate>>, dead: &[u32], n_slots: usize) -> Vec<Vec<sys::tfa_gate>> {
    use std::cmp::Reverse;
    use std::collections::BinaryHeap;
    let gates: Vec<sys::tfa_gate> = levels.iter().flatten().copied().collect(); // level order is a topological order
    let n = gates.len();
    let operands = |g: &sys::tfa_gate| -> Vec<u32> {
        if g.op == sys::TFA_MUX { vec![g.in0, g.in1, g.in2] } else { vec![g.in0, g.in1] }
    };
    let is_xor = |g: &sys::tfa_gate| g.op == sys::TFA_XOR || g.op == sys::TFA_XNOR;
    let mut producer = vec![usize::MAX; n_slots];
    for (i, g) in gates.iter().enumerate() {
        producer[g.out as usize] = i;
    }
    let mut is_dead = vec![false; n_slots];
    for &s in dead {
        is_dead[s as usize] = true;
    }
    let (mut uses, mut user) = (vec![0u32; n], vec![usize::MAX; n]);
    for (i, g) in gates.iter().enumerate() {
        for s in operands(g) {
            let p = producer[s as usize];
            if p != usize::MAX {
                uses[p] += 1;
                user[p] = i;
            }
        }
    }
    let absorbed: Vec<bool> =
        (0..n).map(|i| is_xor(&gates[i]) && is_dead[gates[i].out as usize] && uses[i] == 1 && is_xor(&gates[user[i]])).collect();
    if !absorbed.iter().any(|&a| a) {
        return levels;
    }
    let mut level = vec![0u32; n_slots];
    let mut out: Vec<Vec<sys::tfa_gate>> = Vec::new();
    fn put(out: &mut Vec<Vec<sys::tfa_gate>>, level: &mut [u32], g: sys::tfa_gate, lvl: u32) {
        while out.len() < lvl as usize {
            out.push(Vec::new());
        }
        out[lvl as usize - 1].push(g);
        level[g.out as usize] = lvl;
    }
    for i in 0..n {
        if absorbed[i] {
            continue;
        }
        let g = gates[i];
        let is_root = is_xor(&g) && [g.in0, g.in1].iter().any(|&s| producer[s as usize] != usize::MAX && absorbed[producer[s as usize]]);
        if !is_root {
            let lvl = operands(&g).iter().map(|&s| level[s as usize]).max().unwrap_or(0);
            put(&mut out, &mut level, g, lvl + 1);
            continue;
        }
        let mut heap: BinaryHeap<Reverse<(u32, u32, u32)>> = BinaryHeap::new(); // (level, arrival, slot): earliest first
        let (mut spare, mut stack, mut parity, mut seq) = (Vec::new(), vec![i], 0u32, 0u32);
        while let Some(j) = stack.pop() {
            let x = gates[j];
            parity ^= (x.op == sys::TFA_XNOR) as u32 ^ (x.neg & 1) ^ ((x.neg >> 1) & 1);
            for s in [x.in0, x.in1] {
                let p = producer[s as usize];
                if p != usize::MAX && absorbed[p] {
                    spare.push(s);
                    stack.push(p);
                } else {
                    heap.push(Reverse((level[s as usize], seq, s)));
                    seq += 1;
                }
            }
        }
        while heap.len() > 2 {
            let (Reverse(a), Reverse(b)) = (heap.pop().unwrap(), heap.pop().unwrap());
            let (slot, lvl) = (spare.pop().unwrap(), a.0.max(b.0) + 1);
            put(&mut out, &mut level, sys::tfa_gate { op: sys::TFA_XOR, in0: a.2, in1: b.2, in2: 0, out: slot, neg: 0 }, lvl);
            heap.push(Reverse((lvl, seq, slot)));
            seq += 1;
        }
        let (Reverse(a), Reverse(b)) = (heap.pop().unwrap(), heap.pop().unwrap());
        put(&mut out, &mut level, sys::tfa_gate { op: sys::TFA_XOR, in0: a.2, in1: b.2, in2: 0, out: g.out, neg: parity }, a.0.max(b.0) + 1);
    }
    out
}

struct Engine {
    ctx: *mut sys::tfa_ctx,
    next_fresh: u32,
    capacity: u32,
    free: Vec<u32>,                  // slots whose readers have all been submitted
    released: Arc<Mutex<Vec<u32>>>,  // slots dropped since the last flush: recorded gates may still read them
    recording: bool,
    xor_rebalance: bool,             // re-associate unobservable xor chains before a flush (rebalance_xor_chains)
    levels: Vec<Vec<sys::tfa_gate>>, // levels[l - 1] = the recorded gates of dependency level l
    pending_gates: usize,
    stats: HostStats,
}
unsafe impl Send for Engine {}

impl Engine {
    fn check(&self, rc: i32) {
        if rc != sys::TFA_OK {
            let msg = unsafe { CStr::from_ptr(sys::tfa_last_error(self.ctx)) };
            panic!("tfhe-aes-hip error {rc}: {}", msg.to_string_lossy());
        }
    }
    fn alloc(&mut self) -> u32 {
        if !self.recording {
            // eager: every earlier reader of a released slot is already on the stream, and the stream is ordered
            let mut r = self.released.lock().unwrap();
            self.free.append(&mut r);
        }
        if let Some(s) = self.free.pop() {
            return s;
        }
        if self.next_fresh == self.capacity {
            self.capacity = (self.capacity * 2).max(1 << 17); // 131072 slots: one AES block in flight without a reallocation
            let rc = unsafe { sys::tfa_pool_reserve(self.ctx, self.capacity) };
            self.check(rc);
        }
        self.next_fresh += 1;
        self.next_fresh - 1
    }
    /// Runs the recording: one stage per level, then the slots released meanwhile become reusable.
    fn flush(&mut self, epoch: &AtomicU64) {
        let mut levels = std::mem::take(&mut self.levels);
        if self.xor_rebalance && self.pending_gates > 0 {
            // a slot on the list has no handle left and cannot get one again; one dropped after this snapshot counts as live
            let dead = self.released.lock().unwrap().clone();
            levels = rebalance_xor_chains(levels, &dead, self.next_fresh as usize);
        }
        for lv in levels {
            if lv.is_empty() {
                continue;
            }
            let rc = unsafe { sys::tfa_stage_submit(self.ctx, lv.as_ptr(), lv.len() as u32) };
            self.check(rc);
            self.stats.gates += lv.len() as u64;
            self.stats.stages += 1;
            self.stats.max_stage_width = self.stats.max_stage_width.max(lv.len() as u64);
        }
        let rc = unsafe { sys::tfa_sync(self.ctx) };
        self.check(rc);
        self.pending_gates = 0;
        self.stats.flushes += 1;
        epoch.fetch_add(1, Ordering::SeqCst);
        let mut r = self.released.lock().unwrap();
        self.free.append(&mut r);
    }
}
impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::tfa_ctx_destroy(self.ctx) };
    }
}

/// `tfhe::boolean::ServerKey`: `Clone` and shareable between threads like the original
/// (primitive.rs:21, `&ServerKey` crossing rayon workers); recording and submission are serialised on the context.
#[derive(Clone)]
pub struct ServerKey {
    engine: Arc<Mutex<Engine>>,
    epoch: Arc<AtomicU64>, // number of flushes: a handle written in an older epoch is complete (level 0)
    released: Arc<Mutex<Vec<u32>>>,
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
        let released = Arc::new(Mutex::new(Vec::new()));
        let engine = Engine {
            ctx, next_fresh: 0, capacity: 0, free: Vec::new(), released: released.clone(), recording: true, xor_rebalance: true, levels: Vec::new(), pending_gates: 0,
            stats: HostStats::default(),
        };
        engine.check(sys::tfa_keys_upload(ctx, bsk.as_ptr(), ksk.as_ptr()));
        let server = ServerKey { engine: Arc::new(Mutex::new(engine)), epoch: Arc::new(AtomicU64::new(0)), released };
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
        // a slot on the free list has no reader among the recorded gates, so writing it now is safe in both modes
        let slots: Vec<u32> = (0..bits.len()).map(|_| e.alloc()).collect();
        let rc = unsafe { sys::tfa_pool_write(e.ctx, slots.as_ptr(), slots.len() as u32, lwe.as_ptr()) };
        e.check(rc);
        let epoch = self.server.epoch.load(Ordering::SeqCst);
        slots.into_iter().map(|index| self.server.handle(index, 0, epoch)).collect()
    }
    /// `ClientKey::decrypt` (primitive.rs:85): phase < 2^31.  Runs whatever is recorded first.
    pub fn decrypt(&self, ct: &Ciphertext) -> bool {
        self.decrypt_many(std::slice::from_ref(ct))[0]
    }
    pub fn decrypt_many(&self, cts: &[Ciphertext]) -> Vec<bool> {
        let n1 = self.params.lwe_dimension as usize + 1;
        let enc: Vec<(usize, &Slot)> = cts.iter().enumerate().filter_map(|(i, c)| if let Ciphertext::Encrypted(s) = c { Some((i, s)) } else { None }).collect();
        let mut out: Vec<bool> = cts.iter().map(|c| matches!(c, Ciphertext::Trivial(true))).collect();
        if enc.is_empty() {
            return out;
        }
        let slots: Vec<u32> = enc.iter().map(|(_, s)| s.index()).collect();
        let mut lwe = vec![0u32; slots.len() * n1];
        {
            let mut e = self.server.engine.lock().unwrap();
            e.flush(&self.server.epoch);
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

/// A group of gates none of which reads another's output.  Replaces a rayon `par_iter` over gate calls: record
/// with `gate`/`mux`, then `run`.  On an eager key `run` submits the group as ONE stage; on a recording key it
/// files every gate at its own dependency level, to be submitted by the next flush.
pub struct StageBatch<'a> {
    key: &'a ServerKey,
    pending: Vec<Pending>,
    gates: Vec<sys::tfa_gate>, // `out` holds the ticket until `run` assigns slots
    levels: Vec<u32>,          // per gate: deepest level among its operands, as of the epoch in `epochs`
    epochs: Vec<u64>,
    held: Vec<Arc<SlotRef>>,   // operands stay alive until their gate has been filed
}

impl ServerKey {
    fn handle(&self, index: u32, level: u32, epoch: u64) -> Ciphertext {
        Ciphertext::Encrypted(Slot { handle: Arc::new(SlotRef { index, level, epoch, released: self.released.clone() }), negated: false })
    }
    /// `true` (default): gates are recorded and run level by level at the next flush / decrypt; `false`: every
    /// `StageBatch::run` (and every single gate call) submits at once.  Switching runs what is recorded.
    pub fn set_recording(&self, on: bool) {
        let mut e = self.engine.lock().unwrap();
        e.flush(&self.epoch);
        e.recording = on;
    }
    /// Runs the recorded gates, one `tfa_stage_submit` per dependency level, and waits for them.
    /// On (the default): chains of xor gates whose intermediate values have no live handle are re-associated into trees of
    /// least depth before a flush (`rebalance_xor_chains`); off: every gate runs in the association it was recorded in.
    pub fn set_xor_rebalance(&self, on: bool) {
        self.engine.lock().unwrap().xor_rebalance = on;
    }
    pub fn flush(&self) {
        self.engine.lock().unwrap().flush(&self.epoch);
    }
    pub fn host_stats(&self) -> HostStats {
        self.engine.lock().unwrap().stats
    }
    pub fn trivial_encrypt(&self, bit: bool) -> Ciphertext {
        Ciphertext::Trivial(bit)
    }
    pub fn not(&self, ct: &Ciphertext) -> Ciphertext {
        match ct {
            Ciphertext::Trivial(b) => Ciphertext::Trivial(!b),
            Ciphertext::Encrypted(s) => Ciphertext::Encrypted(Slot { handle: s.handle.clone(), negated: !s.negated }),
        }
    }
    pub fn not_assign(&self, ct: &mut Ciphertext) {
        *ct = self.not(ct);
    }
    pub fn batch(&self) -> StageBatch<'_> {
        StageBatch { key: self, pending: Vec::new(), gates: Vec::new(), levels: Vec::new(), epochs: Vec::new(), held: Vec::new() }
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
    fn file(&mut self, code: u32, ins: &[&Slot]) -> Ticket {
        self.pending.push(Pending::Gate);
        let ticket = self.pending.len() - 1;
        let now = self.key.epoch.load(Ordering::SeqCst);
        let mut g = sys::tfa_gate { op: code, in0: 0, in1: 0, in2: 0, out: ticket as u32, neg: 0 };
        let mut level = 0;
        for (k, s) in ins.iter().enumerate() {
            match k {
                0 => g.in0 = s.index(),
                1 => g.in1 = s.index(),
                _ => g.in2 = s.index(),
            }
            g.neg |= (s.negated as u32) << k;
            if s.handle.epoch == now {
                level = level.max(s.handle.level);
            }
            self.held.push(s.handle.clone());
        }
        self.gates.push(g);
        self.levels.push(level);
        self.epochs.push(now);
        ticket
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
                self.file(code, &[x, y])
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
            (Encrypted(s), Encrypted(x), Encrypted(y)) => self.file(sys::TFA_MUX, &[s, x, y]),
        }
    }
    /// Result k belongs to ticket k.  Eager key: the gates run now, as one stage.  Recording key: each gate is filed
    /// at level 1 + (deepest operand level) of the current recording and runs at the next flush.
    pub fn run(self) -> Vec<Ciphertext> {
        let StageBatch { key, pending, mut gates, levels, epochs, held } = self;
        let mut slot_of: Vec<Option<Ciphertext>> = (0..pending.len()).map(|_| None).collect();
        {
            let mut e = key.engine.lock().unwrap();
            if e.recording && e.pending_gates + gates.len() > AUTO_FLUSH_GATES {
                e.flush(&key.epoch);
            }
            let now = key.epoch.load(Ordering::SeqCst);
            for (k, g) in gates.iter_mut().enumerate() {
                let ticket = g.out as usize;
                g.out = e.alloc();
                // operand levels were read in epoch `epochs[k]`; after a flush in between they are all complete
                let level = if e.recording { 1 + if epochs[k] == now { levels[k] } else { 0 } } else { 0 };
                slot_of[ticket] = Some(key.handle(g.out, level, now));
                if e.recording {
                    if e.levels.len() < level as usize {
                        e.levels.resize(level as usize, Vec::new());
                    }
                    e.levels[level as usize - 1].push(*g);
                    e.pending_gates += 1;
                }
            }
            if !e.recording && !gates.is_empty() {
                let rc = unsafe { sys::tfa_stage_submit(e.ctx, gates.as_ptr(), gates.len() as u32) };
                e.check(rc);
                e.stats.gates += gates.len() as u64;
                e.stats.stages += 1;
                e.stats.max_stage_width = e.stats.max_stage_width.max(gates.len() as u64);
            }
        }
        drop(held); // the operands' gates are filed (or on the stream): their slots may be released from here on
        pending
            .into_iter()
            .enumerate()
            .map(|(k, p)| match p {
                Pending::Ready(ct) => ct,
                Pending::Gate => slot_of[k].take().expect("slot assigned"),
            })
            .collect()
    }
}
