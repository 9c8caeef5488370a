//! `base` of pjanthony2001/tfhe-aes with its gate calls routed to the MI355X engine.
//! `boolean_tree` and `sbox` are the reference's own modules (see README.md); the others are written on `hip`.
pub mod boolean_tree;
pub mod hip;
pub mod key_schedule;
pub mod primitive;
pub mod sbox;
pub mod state;

pub use key_schedule::Key;
pub use primitive::FHEByte;
pub use state::State;
