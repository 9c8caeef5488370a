//! Sketch of the `base` side of the binding: the stage loop of `FHEByte::sub_byte`
//! (base/src/primitive.rs:250-265) with the rayon pool replaced by one `tfa_stage_submit` per stage.
//! Source for a maintainer to adapt; it has not been compiled (no Rust toolchain in the build image).
//!
//! Ciphertexts live in the engine's device pool; the host side holds slot handles.  A handle
//! carries the "negated" flag that `ServerKey::not` sets (the engine folds the negation into the
//! consumer's linear combination), so `Operand::NotBit_k` costs nothing.

use std::collections::HashMap;
use tfhe_aes_hip_sys as sys;

use crate::boolean_tree::{BooleanExpr, Operand};

#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct Slot {
    pub index: u32,
    pub negated: bool,
}

pub struct HipServerKey {
    ctx: *mut sys::tfa_ctx,
    next_slot: u32,
    capacity: u32,
}

impl HipServerKey {
    /// `bsk`/`ksk`: the standard-domain key material of `tfhe::boolean::ServerKey`
    /// ([n][l][k+1][k+1][N] and [kN][l_ks][n+1] u32 words).
    pub fn new(bsk: &[u32], ksk: &[u32], hip_device: i32) -> Result<Self, String> {
        unsafe {
            let mut p = std::mem::zeroed::<sys::tfa_params>();
            sys::tfa_default_params(&mut p);
            let mut ctx = std::ptr::null_mut();
            let rc = sys::tfa_ctx_create(&p, hip_device, &mut ctx);
            let this = Self { ctx, next_slot: 0, capacity: 0 };
            this.check(rc)?;
            this.check(sys::tfa_keys_upload(ctx, bsk.as_ptr(), ksk.as_ptr()))?;
            Ok(this)
        }
    }

    fn check(&self, rc: i32) -> Result<(), String> {
        if rc == sys::TFA_OK {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::tfa_last_error(self.ctx)) };
        Err(format!("tfhe-aes-hip error {rc}: {}", msg.to_string_lossy()))
    }

    fn alloc(&mut self, count: u32) -> Result<u32, String> {
        let first = self.next_slot;
        self.next_slot += count;
        if self.next_slot > self.capacity {
            self.capacity = (self.next_slot * 2).max(1 << 16);
            self.check(unsafe { sys::tfa_pool_reserve(self.ctx, self.capacity) })?;
        }
        Ok(first)
    }

    /// Upload client-encrypted LWE ciphertexts ([count][n+1] u32) and return their slots.
    pub fn import(&mut self, lwe: &[u32], count: u32) -> Result<Vec<Slot>, String> {
        let first = self.alloc(count)?;
        let slots: Vec<u32> = (first..first + count).collect();
        self.check(unsafe { sys::tfa_pool_write(self.ctx, slots.as_ptr(), count, lwe.as_ptr()) })?;
        Ok(slots.into_iter().map(|index| Slot { index, negated: false }).collect())
    }

    pub fn export(&mut self, slots: &[Slot], lwe_out: &mut [u32]) -> Result<(), String> {
        // a negated handle is materialised with a NOT gate before it leaves the pool
        let plain: Vec<u32> = slots.iter().map(|s| s.index).collect();
        self.check(unsafe { sys::tfa_pool_read(self.ctx, plain.as_ptr(), plain.len() as u32, lwe_out.as_mut_ptr()) })
    }

    pub fn not(&self, a: Slot) -> Slot {
        Slot { index: a.index, negated: !a.negated }
    }

    /// One stage of one or many S-box evaluations: every expression of `stage` has its operands in
    /// `operands` (leaves) or `done` (earlier stages).  Replaces
    /// `grouped_by_stage[i] ... into_par_iter().map_with(server_key, |sk, (expr, runnable)| runnable.run(sk))`.
    pub fn run_stage(
        &mut self,
        stage: &[BooleanExpr],
        operands: &HashMap<Operand, Slot>,
        done: &mut HashMap<BooleanExpr, Slot>,
    ) -> Result<(), String> {
        let first = self.alloc(stage.len() as u32)?;
        let mut gates = Vec::with_capacity(stage.len());
        for (k, expr) in stage.iter().enumerate() {
            let out = first + k as u32;
            let arg = |s: Slot, bit: u32| (s.index, if s.negated { 1u32 << bit } else { 0 });
            let gate = match expr {
                BooleanExpr::Operand(op) => {
                    let (a, n) = arg(operands[op], 0);
                    sys::tfa_gate { op: sys::TFA_COPY, in0: a, in1: 0, in2: 0, out, neg: n }
                }
                BooleanExpr::And(x, y) | BooleanExpr::Or(x, y) | BooleanExpr::Xor(x, y) => {
                    let ((a, na), (b, nb)) = (arg(done[x.as_ref()], 0), arg(done[y.as_ref()], 1));
                    let op = match expr {
                        BooleanExpr::And(..) => sys::TFA_AND,
                        BooleanExpr::Or(..) => sys::TFA_OR,
                        _ => sys::TFA_XOR,
                    };
                    sys::tfa_gate { op, in0: a, in1: b, in2: 0, out, neg: na | nb }
                }
                BooleanExpr::Mux(sel, x, y) => {
                    let ((c, nc), (a, na), (b, nb)) = (arg(operands[sel], 0), arg(done[x.as_ref()], 1), arg(done[y.as_ref()], 2));
                    sys::tfa_gate { op: sys::TFA_MUX, in0: c, in1: a, in2: b, out, neg: nc | na | nb }
                }
            };
            gates.push(gate);
        }
        self.check(unsafe { sys::tfa_stage_submit(self.ctx, gates.as_ptr(), gates.len() as u32) })?;
        for (k, expr) in stage.iter().enumerate() {
            done.insert(expr.clone(), Slot { index: first + k as u32, negated: false });
        }
        Ok(())
    }
}

impl Drop for HipServerKey {
    fn drop(&mut self) {
        unsafe {
            sys::tfa_ctx_destroy(self.ctx);
        }
    }
}
