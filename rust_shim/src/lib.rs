//! candle-birefnet with the MI355X backend: the crate root of the reference (src/lib.rs:6-14) with the same public items,
//! implemented over the C ABI of libbirefnet_hip.so instead of candle ops.
//!
//!   pub mod {deform_conv, decoder, aspp, birefnet, swin};
//!   pub use birefnet::BiRefNet;  pub use deform_conv::DeformableConv2d;  pub use swin::{SwinTransformer, SwinConfig};
#![cfg(feature = "hip")]

pub mod hip_ffi;
pub mod deform_conv;
pub mod decoder;
pub mod aspp;
pub mod birefnet;
pub mod swin;

pub use birefnet::BiRefNet;
pub use deform_conv::DeformableConv2d;
pub use swin::{SwinConfig, SwinTransformer};
