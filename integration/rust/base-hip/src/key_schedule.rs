//! `Key` and the AES-128 key schedule (base/src/key_schedule.rs:7-210) on the batched engine: same public names,
//! same transposed layout `data[4 row + col]`, same gates per round (4 S-boxes + 128 xors, :90-131); the rayon
//! loops are replaced by one `eval_circuit` call for the four S-boxes and one `bytes_gate` call per xor layer.
use crate::hip::{ClientKey, Op, ServerKey};
use crate::primitive::{bytes_gate, eval_circuit, s_box_circuit, FHEByte};
use crate::sbox::S_BOX_DATA;

const RCON: [u8; 10] = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36];

#[derive(Clone, Debug)]
pub struct Key {
    pub data: [FHEByte; 16],
}

const SWAPS: [(usize, usize); 6] = [(1, 4), (2, 8), (3, 12), (6, 9), (7, 13), (11, 14)];
fn transposed<T>(mut v: [T; 16]) -> [T; 16] {
    for (a, b) in SWAPS {
        v.swap(a, b);
    }
    v
}
fn u128_bytes(value: u128) -> [u8; 16] {
    value.to_be_bytes()
}

impl Key {
    pub fn from_u128_enc(value: u128, client_key: &ClientKey) -> Self {
        Self::from_u8_enc(&u128_bytes(value), client_key)
    }
    pub fn from_u128_clear(value: u128, server_key: &ServerKey) -> Self {
        Self::from_u8_clear(&u128_bytes(value), server_key)
    }
    pub fn from_u8_enc(data: &[u8; 16], client_key: &ClientKey) -> Self {
        Key { data: transposed(data.map(|v| FHEByte::from_u8_enc(&v, client_key))) }
    }
    pub fn from_u8_clear(data: &[u8; 16], server_key: &ServerKey) -> Self {
        Key { data: transposed(data.map(|v| FHEByte::from_u8_clear(&v, server_key))) }
    }

    /// key_schedule.rs:90-131.  temp = S-box of the last column rotated (transposed indices 7, 11, 15, 3), all four
    /// through ONE `eval_circuit` call (seven stages of 100 / 532 / 508 / 256 / 128 / 64 / 32 gates); the round
    /// constant is a trivial byte, so its xor is free; then column 0 ^= temp and three dependent column layers.
    pub fn generate_next_key_in_place(&mut self, rcon: &u8, server_key: &ServerKey) {
        let sk = server_key;
        let rot: [usize; 4] = [7, 11, 15, 3];
        let refs: Vec<&FHEByte> = rot.iter().map(|&i| &self.data[i]).collect();
        let mut temp = eval_circuit(s_box_circuit(), &refs, sk);
        let round_constant = FHEByte::from_u8_clear(rcon, sk);
        temp[0] = bytes_gate(Op::Xor, &[(&temp[0], &round_constant)], sk).pop().unwrap();
        // column c of row r sits at data[4 r + c]: column 0 takes temp, column c + 1 takes the new column c
        let col0: [usize; 4] = [0, 4, 8, 12];
        let pairs: Vec<(&FHEByte, &FHEByte)> = col0.iter().map(|&i| (&self.data[i], &temp[i / 4])).collect();
        for (&i, b) in col0.iter().zip(bytes_gate(Op::Xor, &pairs, sk)) {
            self.data[i] = b;
        }
        for c in 0..3 {
            let dst: [usize; 4] = [c + 1, c + 5, c + 9, c + 13];
            let pairs: Vec<(&FHEByte, &FHEByte)> = dst.iter().map(|&i| (&self.data[i], &self.data[i - 1])).collect();
            for (&i, b) in dst.iter().zip(bytes_gate(Op::Xor, &pairs, sk)) {
                self.data[i] = b;
            }
        }
    }

    pub fn decrypt_to_u8(&self, client_key: &ClientKey) -> [u8; 16] {
        let out: [u8; 16] = core::array::from_fn(|i| self.data[i].decrypt_to_u8(client_key));
        transposed(out)
    }

    /// key_schedule.rs:162-171: the eleven round keys, each from the one before.
    pub fn generate_round_keys(&self, server_key: &ServerKey) -> [Key; 11] {
        let mut keys = vec![self.clone()];
        for i in 0..10 {
            let mut key = keys[i].clone();
            key.generate_next_key_in_place(&RCON[i], server_key);
            keys.push(key);
        }
        keys.try_into().ok().expect("There should be 11 Keys")
    }
}

fn sub_word(word: &[u8; 4]) -> [u8; 4] {
    word.map(|x| S_BOX_DATA[x as usize])
}
fn rot_word(word: &[u8; 4]) -> [u8; 4] {
    [word[1], word[2], word[3], word[0]]
}

/// key_schedule.rs:183-210: the key expansion in the clear.
pub fn key_expansion_clear(key: &[u8; 16]) -> [[u8; 16]; 11] {
    let mut round_keys = [[0u8; 16]; 11];
    round_keys[0] = *key;
    for round in 1..=10 {
        let prev = round_keys[round - 1];
        let mut temp = sub_word(&rot_word(&[prev[12], prev[13], prev[14], prev[15]]));
        temp[0] ^= RCON[round - 1];
        for i in 0..4 {
            round_keys[round][i] = prev[i] ^ temp[i];
        }
        for i in 4..16 {
            round_keys[round][i] = prev[i] ^ round_keys[round][i - 4];
        }
    }
    round_keys
}
