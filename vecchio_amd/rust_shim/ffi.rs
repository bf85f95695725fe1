// ffi.rs — Rust-side binding of include/vecchio_amd.h for browserdotsys/vecchio.
//
// UNCOMPILED SOURCE: there is no Rust toolchain in the build image (no cargo/rustc), so this
// file has never been through rustc.  It is the binding a vecchio maintainer would add as
// `src/ffi.rs` (+ `mod ffi;` in main.rs) together with the `flatten()` methods sketched in
// flatten.rs; INTEGRATION.md walks through the three edits.  The C++ twin of this code that IS
// compiled and tested lives in vecchio_amd/host/ (same structure, same record layout).
#![allow(non_camel_case_types, dead_code)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub type vk_ref = u32;
pub const VK_ABI_VERSION: u32 = 6;
pub const VK_SCENE_FAST_ACCEL: u32 = 1;
pub const VK_SCENE_REFERENCE_TREE: u32 = 2;
pub const VK_SCENE_EMPIRICAL_TREES: u32 = 4;
pub const VK_SCENE_RCCL_GATHER: u32 = 8;
pub const VK_REF_FLIP: u32 = 0x0800_0000;
pub const VK_KIND_BVH: u32 = 1;
pub const VK_KIND_SPHERE: u32 = 2;
pub const VK_KIND_MOVING_SPHERE: u32 = 3;
pub const VK_KIND_RECT: u32 = 4;
pub const VK_KIND_LIST: u32 = 5;
pub const VK_KIND_MEDIUM: u32 = 6;
pub const VK_KIND_TRANSLATE: u32 = 7;
pub const VK_KIND_ROTATE: u32 = 8;
pub fn make_ref(kind: u32, index: usize) -> vk_ref { (kind << 28) | (index as u32 & 0x07FF_FFFF) }

#[repr(C)] #[derive(Copy, Clone)] pub struct vk_bvh_node { pub bb_min: [f32; 3], pub bb_max: [f32; 3], pub left: vk_ref, pub right: vk_ref }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_sphere { pub center: [f32; 3], pub radius: f32, pub material: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_moving_sphere { pub center0: [f32; 3], pub center1: [f32; 3], pub time0: f32, pub time1: f32, pub radius: f32, pub material: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_rect { pub c0: f32, pub c1: f32, pub d0: f32, pub d1: f32, pub k: f32, pub axis0: u8, pub axis1: u8, pub axis2: u8, pub _pad: u8, pub material: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_list { pub first: u32, pub count: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_medium { pub boundary: vk_ref, pub neg_inv_density: f32, pub material: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_translate { pub child: vk_ref, pub offset: [f32; 3] }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_rotate { pub child: vk_ref, pub axis: u32, pub sin_theta: f32, pub cos_theta: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_material { pub kind: u32, pub texture: u32, pub param: f32, pub a: u32, pub b: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_texture { pub kind: u32, pub color: [f32; 3], pub a: u32, pub b: u32, pub scale: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_image { pub width: u32, pub height: u32, pub rgb: *const u8 }
#[repr(C)] #[derive(Copy, Clone)] pub struct vk_perlin { pub ranvec: [[f32; 3]; 256], pub perm_x: [u32; 256], pub perm_y: [u32; 256], pub perm_z: [u32; 256] }

#[repr(C)]
pub struct vk_scene_desc {
    pub abi_version: u32,
    pub n_bvh: u32, pub bvh: *const vk_bvh_node,
    pub n_spheres: u32, pub spheres: *const vk_sphere,
    pub n_moving_spheres: u32, pub moving_spheres: *const vk_moving_sphere,
    pub n_rects: u32, pub rects: *const vk_rect,
    pub n_lists: u32, pub lists: *const vk_list,
    pub n_list_items: u32, pub list_items: *const vk_ref,
    pub n_media: u32, pub media: *const vk_medium,
    pub n_translates: u32, pub translates: *const vk_translate,
    pub n_rotates: u32, pub rotates: *const vk_rotate,
    pub n_materials: u32, pub materials: *const vk_material,
    pub n_textures: u32, pub textures: *const vk_texture,
    pub n_images: u32, pub images: *const vk_image,
    pub n_perlins: u32, pub perlins: *const vk_perlin,
    pub world: vk_ref,
    pub n_lights: u32, pub lights: *const vk_ref,
    pub flags: u32,             // 0 = the handed-over tree's results (exact re-treeing of sphere-only worlds, see vecchio_amd.h); VK_SCENE_*
}

#[repr(C)] #[derive(Copy, Clone)]
pub struct vk_camera {          // the ten fields of main.rs:57-68 (add #[repr(C)] there, or copy)
    pub origin: [f32; 3], pub lower_left_corner: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3],
    pub u: [f32; 3], pub v: [f32; 3], pub w: [f32; 3], pub lens_radius: f32, pub time0: f32, pub time1: f32,
}

#[repr(C)] #[derive(Copy, Clone)]
pub struct vk_render_params {
    pub width: u32, pub height: u32, pub samples_per_pixel: u32, pub max_depth: u32, pub seed: u64,
    pub integrator: u32, pub background: u32, pub background_color: [f32; 3], pub tile_rank: u32, pub tile_world: u32,
    pub output_format: u32,     // VK_OUTPUT_F32 = 0 | VK_OUTPUT_RGB8 = 1 (Vec3::to_color + top-down rows fused, vec3.rs:54-61, main.rs:209)
}

#[repr(C)] #[derive(Copy, Clone, Default)]
pub struct vk_stats { pub samples: u64, pub seconds: f64, pub kernel_ms: f64, pub kernel_launches: u32, pub scene_in_lds: u32, pub clamped_samples: u64 }

#[repr(C)] pub struct vk_scene { _private: [u8; 0] }

#[link(name = "vecchio_amd")]
extern "C" {
    pub fn vk_abi_version() -> c_int;
    pub fn vk_device_count() -> c_int;
    pub fn vk_last_error() -> *const c_char;
    pub fn vk_scene_create(desc: *const vk_scene_desc, device: c_int, out: *mut *mut vk_scene) -> c_int;
    pub fn vk_scene_create_multi(desc: *const vk_scene_desc, devices: *const c_int, n_devices: c_int, out: *mut *mut vk_scene) -> c_int;
    pub fn vk_scene_destroy(scene: *mut vk_scene);
    pub fn vk_render(scene: *mut vk_scene, cam: *const vk_camera, params: *const vk_render_params, rgb_out: *mut f32, stats: *mut vk_stats) -> c_int;
    pub fn vk_render_device(scene: *mut vk_scene, cam: *const vk_camera, params: *const vk_render_params, d_rgb: *mut c_void, stream: *mut c_void, stats: *mut vk_stats) -> c_int;
    pub fn vk_scene_last_kernel_ms(scene: *mut vk_scene, ms_out: *mut f64) -> c_int;
    pub fn vk_scene_last_clamped_samples(scene: *mut vk_scene, count_out: *mut u64) -> c_int;
    pub fn vk_scene_last_requeued_samples(scene: *mut vk_scene, count_out: *mut u64) -> c_int;
    pub fn vk_tile_slab_bytes(width: u32, height: u32, output_format: u32, tile_rank: u32, tile_world: u32) -> usize;
    pub fn vk_pack_tiles_device(scene: *mut vk_scene, d_fb: *const c_void, width: u32, height: u32, output_format: u32, tile_rank: u32, tile_world: u32, d_slab: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn vk_unpack_tiles_device(scene: *mut vk_scene, d_slab: *const c_void, width: u32, height: u32, output_format: u32, tile_rank: u32, tile_world: u32, d_img: *mut c_void, stream: *mut c_void) -> c_int;
}

/// What `flatten()` pushes into (flatten.rs).  One record per Arc; shared Arcs are de-duplicated
/// by pointer identity so a light that is both in `world` and in `lights` flattens once.
#[derive(Default)]
pub struct FlatBuilder {
    pub bvh: Vec<vk_bvh_node>, pub spheres: Vec<vk_sphere>, pub moving_spheres: Vec<vk_moving_sphere>, pub rects: Vec<vk_rect>,
    pub lists: Vec<vk_list>, pub list_items: Vec<vk_ref>, pub media: Vec<vk_medium>, pub translates: Vec<vk_translate>,
    pub rotates: Vec<vk_rotate>, pub materials: Vec<vk_material>, pub textures: Vec<vk_texture>, pub images: Vec<vk_image>,
    pub perlins: Vec<vk_perlin>, pub lights: Vec<vk_ref>, pub world: vk_ref,
    pub seen: std::collections::HashMap<usize, u32>,
}

impl FlatBuilder {
    pub fn desc(&self) -> vk_scene_desc {
        vk_scene_desc {
            abi_version: VK_ABI_VERSION,
            n_bvh: self.bvh.len() as u32, bvh: self.bvh.as_ptr(),
            n_spheres: self.spheres.len() as u32, spheres: self.spheres.as_ptr(),
            n_moving_spheres: self.moving_spheres.len() as u32, moving_spheres: self.moving_spheres.as_ptr(),
            n_rects: self.rects.len() as u32, rects: self.rects.as_ptr(),
            n_lists: self.lists.len() as u32, lists: self.lists.as_ptr(),
            n_list_items: self.list_items.len() as u32, list_items: self.list_items.as_ptr(),
            n_media: self.media.len() as u32, media: self.media.as_ptr(),
            n_translates: self.translates.len() as u32, translates: self.translates.as_ptr(),
            n_rotates: self.rotates.len() as u32, rotates: self.rotates.as_ptr(),
            n_materials: self.materials.len() as u32, materials: self.materials.as_ptr(),
            n_textures: self.textures.len() as u32, textures: self.textures.as_ptr(),
            n_images: self.images.len() as u32, images: self.images.as_ptr(),
            n_perlins: self.perlins.len() as u32, perlins: self.perlins.as_ptr(),
            world: self.world, n_lights: self.lights.len() as u32, lights: self.lights.as_ptr(), flags: 0,
        }
    }
}

/// Owns the uploaded scene; `render` is the drop-in for the closure at main.rs:181-198.
pub struct GpuScene { handle: *mut vk_scene }

fn check(status: c_int) -> Result<(), std::io::Error> {
    if status == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(vk_last_error()) }.to_string_lossy().into_owned();
    Err(std::io::Error::new(std::io::ErrorKind::Other, format!("vecchio_amd status {}: {}", status, msg)))
}

impl GpuScene {
    pub fn new(fb: &FlatBuilder, device: i32) -> Result<GpuScene, std::io::Error> {
        let mut h: *mut vk_scene = std::ptr::null_mut();
        let d = fb.desc();
        check(unsafe { vk_scene_create(&d, device, &mut h) })?;
        Ok(GpuScene { handle: h })
    }
    /// One handle over every listed GPU: `render` then deals the tiles over them and gathers on devices[0]
    /// (what `main()`'s single-threaded frame loop, main.rs:176, calls on an 8-GPU node).
    pub fn new_multi(fb: &FlatBuilder, devices: &[i32]) -> Result<GpuScene, std::io::Error> {
        let mut h: *mut vk_scene = std::ptr::null_mut();
        let d = fb.desc();
        check(unsafe { vk_scene_create_multi(&d, devices.as_ptr(), devices.len() as c_int, &mut h) })?;
        Ok(GpuScene { handle: h })
    }
    /// `pixels`: width*height Vec3 (#[repr(C)] added to vec3.rs:3-8), y = 0 bottom row as in main.rs:182-183.
    pub fn render(&self, cam: &vk_camera, params: &vk_render_params, pixels: &mut [[f32; 3]]) -> Result<vk_stats, std::io::Error> {
        assert_eq!(pixels.len(), (params.width * params.height) as usize);
        let mut st = vk_stats::default();
        check(unsafe { vk_render(self.handle, cam, params, pixels.as_mut_ptr() as *mut f32, &mut st) })?;
        Ok(st)
    }
}
impl Drop for GpuScene { fn drop(&mut self) { unsafe { vk_scene_destroy(self.handle) } } }
