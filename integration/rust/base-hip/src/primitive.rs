//! `FHEByte` (base/src/primitive.rs:50-368) on the batched engine.  Same public names and bit order
//! (`data[0]` is the most significant bit, primitive.rs:68-74); the rayon loops over gates are replaced by stage
//! batches (`hip::StageBatch`), and every operation has a `*_many` form that puts the gates of many bytes into
//! the same stages, which is what `State` uses.
use std::collections::{HashMap, HashSet, VecDeque};
use std::sync::OnceLock;

use crate::boolean_tree::{BooleanExpr, Operand};
use crate::hip::{Ciphertext, ClientKey, Op, ServerKey};
use crate::sbox::{generate_reduced_bool_expr, INV_S_BOX_DATA, S_BOX_DATA};

/// A circuit staged once per process: `roots` MSB first (sbox.rs:53-61), `stages[s]` = the unique
/// sub-expressions of height s in the total order of boolean_tree.rs:519-562.  The reference rebuilds the
/// hash set on every call (primitive.rs:212-225); the staging never changes, so it is kept.
pub struct StagedCircuit {
    pub roots: Vec<BooleanExpr>,
    pub stages: Vec<Vec<BooleanExpr>>,
}

impl StagedCircuit {
    pub fn new(roots: Vec<BooleanExpr>) -> Self {
        let mut nodes: HashSet<BooleanExpr> = HashSet::new();
        for r in &roots {
            r.to_hashset(&mut nodes);
        }
        let depth = nodes.iter().map(|e| e.stage() as usize).max().unwrap_or(0);
        let mut stages: Vec<Vec<BooleanExpr>> = vec![Vec::new(); depth + 1];
        for e in nodes {
            let s = e.stage() as usize;
            stages[s].push(e);
        }
        for s in &mut stages {
            s.sort(); // a fixed gate order inside a stage: the submitted lists are reproducible
        }
        StagedCircuit { roots, stages }
    }
}

pub fn s_box_circuit() -> &'static StagedCircuit {
    static C: OnceLock<StagedCircuit> = OnceLock::new();
    C.get_or_init(|| StagedCircuit::new(generate_reduced_bool_expr(S_BOX_DATA)))
}
pub fn inv_s_box_circuit() -> &'static StagedCircuit {
    static C: OnceLock<StagedCircuit> = OnceLock::new();
    C.get_or_init(|| StagedCircuit::new(generate_reduced_bool_expr(INV_S_BOX_DATA)))
}

#[derive(Clone, Debug)]
pub struct FHEByte {
    pub(crate) data: VecDeque<Ciphertext>,
}

/// Operand -> ciphertext of one byte (primitive.rs:228-248): Bit_k is bit k counted from the least significant
/// end, NotBit_k its free negation, True/False trivial.
fn leaf(byte: &FHEByte, op: &Operand, sk: &ServerKey) -> Ciphertext {
    let code = *op as u8; // True = 0, False = 1, Bit_k = 2 + 2k, NotBit_k = 3 + 2k (boolean_tree.rs:15-34)
    if code < 2 {
        return sk.trivial_encrypt(code == 0);
    }
    let k = ((code - 2) / 2) as usize;
    let bit = &byte.data[7 - k];
    if code % 2 == 0 { bit.clone() } else { sk.not(bit) }
}

/// The staged evaluation (primitive.rs:250-265 and its twin :318-333) for any number of bytes: stage s of every
/// byte goes into ONE `tfa_stage_submit`.  Gate order inside a stage: byte-major, expressions in `stages[s]` order.
pub fn eval_circuit(circuit: &StagedCircuit, bytes: &[&FHEByte], sk: &ServerKey) -> Vec<FHEByte> {
    let mut done: Vec<HashMap<&BooleanExpr, Ciphertext>> = vec![HashMap::new(); bytes.len()];
    // height 0: the operands themselves (the `Operand(_) => clone` arm of Runnable::run, boolean_tree.rs:646)
    for (b, byte) in bytes.iter().enumerate() {
        for e in &circuit.stages[0] {
            if let BooleanExpr::Operand(op) = e {
                done[b].insert(e, leaf(byte, op, sk));
            }
        }
    }
    for stage in circuit.stages.iter().skip(1) {
        let mut batch = sk.batch();
        let mut tickets = Vec::with_capacity(stage.len() * bytes.len());
        for (b, byte) in bytes.iter().enumerate() {
            for e in stage {
                let t = match e {
                    BooleanExpr::And(x, y) => batch.gate(Op::And, &done[b][x.as_ref()], &done[b][y.as_ref()]),
                    BooleanExpr::Or(x, y) => batch.gate(Op::Or, &done[b][x.as_ref()], &done[b][y.as_ref()]),
                    BooleanExpr::Xor(x, y) => batch.gate(Op::Xor, &done[b][x.as_ref()], &done[b][y.as_ref()]),
                    BooleanExpr::Mux(sel, x, y) => batch.mux(&leaf(byte, sel, sk), &done[b][x.as_ref()], &done[b][y.as_ref()]),
                    BooleanExpr::Operand(_) => unreachable!("operands have height 0"),
                };
                tickets.push((b, e, t));
            }
        }
        let results = batch.run();
        for (b, e, t) in tickets {
            done[b].insert(e, results[t].clone());
        }
    }
    (0..bytes.len())
        .map(|b| FHEByte { data: circuit.roots.iter().map(|r| done[b][r].clone()).collect() })
        .collect()
}

/// One two-input gate on every bit of every pair: a single stage of 8 * pairs.len() gates
/// (the `par_iter_mut().zip(..)` chains of state.rs:72-254 around primitive.rs:96-133).
pub fn bytes_gate(op: Op, pairs: &[(&FHEByte, &FHEByte)], sk: &ServerKey) -> Vec<FHEByte> {
    let mut batch = sk.batch();
    let tickets: Vec<Vec<usize>> = pairs.iter().map(|(x, y)| x.data.iter().zip(y.data.iter()).map(|(a, b)| batch.gate(op, a, b)).collect()).collect();
    let results = batch.run();
    tickets.into_iter().map(|ts| FHEByte { data: ts.into_iter().map(|t| results[t].clone()).collect() }).collect()
}

/// Multiplication by x in GF(2^8) (primitive.rs:349-362) for many bytes in one stage: c = MSB, shift left with a
/// trivial false, then per bit mux(c, x ^ bit_of(0x1b), x).  The xor with a trivial bit is a free negation and the
/// mux on the trivial LSB resolves on the host, so a byte costs 7 MUX gates.
pub fn xtime_many(bytes: &[&FHEByte], sk: &ServerKey) -> Vec<FHEByte> {
    let mut batch = sk.batch();
    let mut tickets = Vec::with_capacity(bytes.len());
    for byte in bytes {
        let c = byte.data[0].clone();
        let mut shifted: Vec<Ciphertext> = byte.data.iter().skip(1).cloned().collect();
        shifted.push(sk.trivial_encrypt(false));
        let ts: Vec<usize> = shifted
            .iter()
            .enumerate()
            .map(|(i, x)| {
                let poly_bit = 0x1bu8 & (1 << (7 - i)) != 0;
                let flipped = if poly_bit { sk.not(x) } else { x.clone() };
                batch.mux(&c, &flipped, x)
            })
            .collect();
        tickets.push(ts);
    }
    let results = batch.run();
    tickets.into_iter().map(|ts| FHEByte { data: ts.into_iter().map(|t| results[t].clone()).collect() }).collect()
}

impl FHEByte {
    pub fn new(bool_data: &[bool], client_key: &ClientKey) -> Self {
        assert!(bool_data.len() == 8, "FHEByte has to be initialized with 8 booleans !");
        FHEByte { data: client_key.encrypt_many(bool_data).into() }
    }
    pub fn from_u8_enc(value: &u8, client_key: &ClientKey) -> Self {
        let bits: Vec<bool> = (0..8).rev().map(|i| value & (1 << i) != 0).collect();
        FHEByte { data: client_key.encrypt_many(&bits).into() }
    }
    pub fn from_u8_clear(value: &u8, server_key: &ServerKey) -> Self {
        Self::trivial_clear(*value, server_key)
    }
    pub fn trivial_clear(clear_value: u8, server_key: &ServerKey) -> Self {
        FHEByte { data: (0..8).rev().map(|i| server_key.trivial_encrypt(clear_value & (1 << i) != 0)).collect() }
    }
    pub fn trivial_false(server_key: &ServerKey) -> Self {
        Self::trivial_clear(0, server_key)
    }
    pub fn decrypt(&self, client_key: &ClientKey) -> Vec<bool> {
        let bits: Vec<Ciphertext> = self.data.iter().cloned().collect();
        client_key.decrypt_many(&bits)
    }
    pub fn decrypt_to_u8(&self, client_key: &ClientKey) -> u8 {
        self.decrypt(client_key).iter().enumerate().map(|(i, &b)| (b as u8) << (7 - i)).sum()
    }

    pub fn xor_in_place(&mut self, rhs: &Self, server_key: &ServerKey) {
        *self = bytes_gate(Op::Xor, &[(&*self, rhs)], server_key).pop().unwrap();
    }
    pub fn xor(&self, rhs: &Self, server_key: &ServerKey) -> Self {
        bytes_gate(Op::Xor, &[(&*self, rhs)], server_key).pop().unwrap()
    }
    pub fn and_in_place(&mut self, rhs: &Self, server_key: &ServerKey) {
        *self = bytes_gate(Op::And, &[(&*self, rhs)], server_key).pop().unwrap();
    }
    pub fn and(&self, rhs: &Self, server_key: &ServerKey) -> Self {
        bytes_gate(Op::And, &[(&*self, rhs)], server_key).pop().unwrap()
    }
    pub fn or_in_place(&mut self, rhs: &Self, server_key: &ServerKey) {
        *self = bytes_gate(Op::Or, &[(&*self, rhs)], server_key).pop().unwrap();
    }
    pub fn or(&self, rhs: &Self, server_key: &ServerKey) -> Self {
        bytes_gate(Op::Or, &[(&*self, rhs)], server_key).pop().unwrap()
    }
    pub fn not_in_place(&mut self, server_key: &ServerKey) {
        for x in self.data.iter_mut() {
            server_key.not_assign(x);
        }
    }
    pub fn not(&self, server_key: &ServerKey) -> Self {
        let mut r = self.clone();
        r.not_in_place(server_key);
        r
    }

    pub fn sub_byte(&self, server_key: &ServerKey) -> Self {
        eval_circuit(s_box_circuit(), &[&*self], server_key).pop().unwrap()
    }
    pub fn inv_sub_byte(&self, server_key: &ServerKey) -> Self {
        eval_circuit(inv_s_box_circuit(), &[&*self], server_key).pop().unwrap()
    }
    pub fn mul_x_gf2_in_place(&mut self, server_key: &ServerKey) {
        *self = xtime_many(&[&*self], server_key).pop().unwrap();
    }
    pub fn mul_x_gf2(&self, server_key: &ServerKey) -> Self {
        xtime_many(&[&*self], server_key).pop().unwrap()
    }
}
