//! `State` (base/src/state.rs:9-313) on the batched engine: the transposed 4 x 4 byte matrix `data[4 row + col]`,
//! same methods, every layer of independent gates one stage.  The gate sequence (which xors, in which order) is the
//! reference's, so a block costs the same gates: AddRoundKey 128 xors, MixColumns 512 xors + 112 muxes.
use crate::hip::{ClientKey, Op, ServerKey};
use crate::primitive::{bytes_gate, eval_circuit, inv_s_box_circuit, s_box_circuit, xtime_many, FHEByte};
use crate::Key;

#[derive(Clone)]
pub struct State {
    pub(crate) data: [FHEByte; 16],
}

const SWAPS: [(usize, usize); 6] = [(1, 4), (2, 8), (3, 12), (6, 9), (7, 13), (11, 14)];
fn transpose<T>(v: &mut [T; 16]) {
    for (a, b) in SWAPS {
        v.swap(a, b);
    }
}
fn to_array(v: Vec<FHEByte>) -> [FHEByte; 16] {
    v.try_into().ok().expect("16 bytes")
}

impl State {
    pub fn from_u8_enc(data: &[u8; 16], client_key: &ClientKey) -> Self {
        let mut bytes = to_array(data.iter().map(|v| FHEByte::from_u8_enc(v, client_key)).collect());
        transpose(&mut bytes);
        State { data: bytes }
    }
    pub fn from_u128_enc(value: u128, client_key: &ClientKey) -> Self {
        Self::from_u8_enc(&value.to_be_bytes(), client_key)
    }

    /// state.rs:52-60: all sixteen bytes share the seven stages (7 submits of 400 / 2128 / 2032 / 1024 / 512 / 256 /
    /// 128 gates instead of 16 x 7 submits of 25 ... 8).
    pub fn sub_bytes(&mut self, server_key: &ServerKey) {
        let refs: Vec<&FHEByte> = self.data.iter().collect();
        self.data = to_array(eval_circuit(s_box_circuit(), &refs, server_key));
    }
    pub fn inv_sub_bytes(&mut self, server_key: &ServerKey) {
        let refs: Vec<&FHEByte> = self.data.iter().collect();
        self.data = to_array(eval_circuit(inv_s_box_circuit(), &refs, server_key));
    }

    fn xor_layer(lhs: &[FHEByte; 16], rhs_index: &[usize; 16], rhs: &[FHEByte], sk: &ServerKey) -> [FHEByte; 16] {
        let pairs: Vec<(&FHEByte, &FHEByte)> = lhs.iter().zip(rhs_index.iter().map(|&i| &rhs[i])).collect();
        to_array(bytes_gate(Op::Xor, &pairs, sk))
    }

    /// state.rs:72-118.  Five dependent layers: two xor layers on the rows, xtime, two more xor layers.
    pub fn mix_columns(&mut self, server_key: &ServerKey) {
        let d = &self.data;
        let first: [usize; 16] = [8, 9, 10, 11, 8, 9, 10, 11, 0, 1, 2, 3, 0, 1, 2, 3];
        let pairs: Vec<(&FHEByte, &FHEByte)> = first.iter().map(|&i| (&d[i], &d[4 + i])).collect();
        let y = to_array(bytes_gate(Op::Xor, &pairs, server_key));
        let second: [usize; 16] = [4, 5, 6, 7, 0, 1, 2, 3, 12, 13, 14, 15, 8, 9, 10, 11];
        let y = Self::xor_layer(&y, &second, d, server_key);
        let refs: Vec<&FHEByte> = d.iter().collect();
        let doubled = xtime_many(&refs, server_key);
        let same: [usize; 16] = core::array::from_fn(|i| i);
        let y = Self::xor_layer(&y, &same, &doubled, server_key);
        let next_row: [usize; 16] = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3];
        self.data = Self::xor_layer(&y, &next_row, &doubled, server_key);
    }

    /// state.rs:120-215, layer by layer.
    pub fn inv_mix_columns(&mut self, server_key: &ServerKey) {
        let sk = server_key;
        let d = self.data.clone();
        let pairs: Vec<(&FHEByte, &FHEByte)> = (0..16).map(|i| (&d[i], &d[(i + 4) % 16])).collect();
        let y = to_array(bytes_gate(Op::Xor, &pairs, sk)); // x_i ^ x_(i+1)
        let pairs: Vec<(&FHEByte, &FHEByte)> = (0..8).map(|i| (&d[i], &d[i + 8])).collect();
        let mut t0 = bytes_gate(Op::Xor, &pairs, sk); // x_i ^ x_(i+2)
        for _ in 0..2 {
            t0 = xtime_many(&t0.iter().collect::<Vec<_>>(), sk); // 04 (x_i ^ x_(i+2))
        }
        let pairs: Vec<(&FHEByte, &FHEByte)> = (0..4).map(|i| (&y[i], &y[8 + i])).collect();
        let mut t1 = bytes_gate(Op::Xor, &pairs, sk); // x_0 ^ x_1 ^ x_2 ^ x_3
        let y = to_array(xtime_many(&y.iter().collect::<Vec<_>>(), sk));
        let t0_index: [usize; 16] = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7];
        let y = Self::xor_layer(&y, &t0_index, &t0, sk);
        let same: [usize; 16] = core::array::from_fn(|i| i);
        let cols: [usize; 16] = core::array::from_fn(|i| i % 4);
        let x = Self::xor_layer(&d, &same, &y, sk);
        let x = Self::xor_layer(&x, &cols, &t1, sk);
        for _ in 0..3 {
            t1 = xtime_many(&t1.iter().collect::<Vec<_>>(), sk); // 08 (x_0 ^ x_1 ^ x_2 ^ x_3)
        }
        self.data = Self::xor_layer(&x, &cols, &t1, sk);
    }

    pub fn xor_key_enc(&mut self, key: &Key, server_key: &ServerKey) {
        let same: [usize; 16] = core::array::from_fn(|i| i);
        self.data = Self::xor_layer(&self.data, &same, &key.data, server_key);
    }
    /// A clear key: sixteen trivial bytes, so every xor is a free negation or a copy (no gate is submitted).
    pub fn xor_key_clear(&mut self, key: &[u8; 16], server_key: &ServerKey) {
        let mut k = *key;
        transpose(&mut k);
        let bytes: Vec<FHEByte> = k.iter().map(|&v| FHEByte::trivial_clear(v, server_key)).collect();
        let same: [usize; 16] = core::array::from_fn(|i| i);
        self.data = Self::xor_layer(&self.data, &same, &bytes, server_key);
    }
    pub fn xor_state(&mut self, state: &State, server_key: &ServerKey) {
        let same: [usize; 16] = core::array::from_fn(|i| i);
        self.data = Self::xor_layer(&self.data, &same, &state.data, server_key);
    }

    pub fn decrypt_to_u8(&self, client_key: &ClientKey) -> [u8; 16] {
        let mut out: [u8; 16] = core::array::from_fn(|i| self.data[i].decrypt_to_u8(client_key));
        transpose(&mut out);
        out
    }
    pub fn decrypt_to_u128(&self, client_key: &ClientKey) -> u128 {
        u128::from_be_bytes(self.decrypt_to_u8(client_key))
    }

    pub fn shift_rows(&mut self) {
        for row in 1..4 {
            self.data[4 * row..4 * row + 4].rotate_left(row);
        }
    }
    pub fn inv_shift_rows(&mut self) {
        for row in 1..4 {
            self.data[4 * row..4 * row + 4].rotate_right(row);
        }
    }
}
