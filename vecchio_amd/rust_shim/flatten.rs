// flatten.rs — the ONE method the shim adds to each of vecchio's traits, because they offer no
// introspection (hittable.rs:33-42, material.rs:20-41,228-230) and the concrete types' fields are
// module-private: each `impl` block below is pasted next to its type (hittable.rs / accel.rs /
// material.rs / scene.rs), and the three trait declarations gain the one-line method shown first.
//
// UNCOMPILED SOURCE (no Rust toolchain in the build image: never been through rustc).  The compiled,
// tested twin is vecchio_amd/host/host.cpp (`X::flatten(FlatBuilder&)`), the same logic type by type.
use crate::accel::BVHNode;
use crate::ffi::*;
use crate::hittable::*;
use crate::material::*;
use std::sync::Arc;

// ---- added to the traits ------------------------------------------------------------------------
//   trait Hittable { ...; fn flatten(&self, b: &mut FlatBuilder) -> vk_ref; }     // hittable.rs:33-42
//   trait Material { ...; fn flatten(&self, b: &mut FlatBuilder) -> u32; }        // material.rs:20-41
//   trait Texture  { ...; fn flatten(&self, b: &mut FlatBuilder) -> u32; }        // material.rs:228-230

// Arc identity -> record, so Arc::clone'd objects (a light that is in `world` and in `lights`, a material
// shared by a thousand spheres) flatten once.  One table: the three kinds of Arc never share an address.
fn key<T: ?Sized>(a: &Arc<T>) -> usize { Arc::as_ptr(a) as *const u8 as usize }
pub fn hittable(b: &mut FlatBuilder, h: &Arc<HittableSS>) -> vk_ref {
    if let Some(r) = b.seen.get(&key(h)) { return *r; }
    let r = h.flatten(b);
    b.seen.insert(key(h), r);
    r
}
pub fn material(b: &mut FlatBuilder, m: &Arc<MaterialSS>) -> u32 {
    if let Some(r) = b.seen.get(&key(m)) { return *r; }
    let r = m.flatten(b);
    b.seen.insert(key(m), r);
    r
}
pub fn texture(b: &mut FlatBuilder, t: &Arc<TextureSS>) -> u32 {
    if let Some(r) = b.seen.get(&key(t)) { return *r; }
    let r = t.flatten(b);
    b.seen.insert(key(t), r);
    r
}
fn v3(v: crate::vec3::Vec3) -> [f32; 3] { [v.x, v.y, v.z] }
fn list(b: &mut FlatBuilder, items: &[Arc<HittableSS>]) -> vk_ref {          // Boxy::sides and any Vec<Arc<..>>
    let refs: Vec<vk_ref> = items.iter().map(|h| hittable(b, h)).collect();   // children first: they may push lists themselves
    let first = b.list_items.len() as u32;
    b.list_items.extend(refs);
    b.lists.push(vk_list { first, count: items.len() as u32 });
    make_ref(VK_KIND_LIST, b.lists.len() - 1)
}

// ---- hittable.rs ----------------------------------------------------------------------------------
impl Sphere {        // inside `impl Hittable for Sphere` (hittable.rs:64)
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let material = material(b, &self.material);
        b.spheres.push(vk_sphere { center: v3(self.center), radius: self.radius, material });
        make_ref(VK_KIND_SPHERE, b.spheres.len() - 1)
    }
}
impl MovingSphere {  // hittable.rs:152
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let material = material(b, &self.material);
        b.moving_spheres.push(vk_moving_sphere { center0: v3(self.center0), center1: v3(self.center1), time0: self.time0,
                                                 time1: self.time1, radius: self.radius, material });
        make_ref(VK_KIND_MOVING_SPHERE, b.moving_spheres.len() - 1)
    }
}
impl Rect {          // hittable.rs:228
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let material = material(b, &self.mat);
        b.rects.push(vk_rect { c0: self.c0, c1: self.c1, d0: self.d0, d1: self.d1, k: self.k, axis0: self.axis0 as u8,
                               axis1: self.axis1 as u8, axis2: self.axis2 as u8, _pad: 0, material });
        make_ref(VK_KIND_RECT, b.rects.len() - 1)
    }
}
impl FlipFace {      // hittable.rs:298 — only negates `front` (hittable.rs:299-308): one bit of the reference, no record
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref { hittable(b, &self.ptr) ^ VK_REF_FLIP }
}
impl Boxy {          // hittable.rs:361 — hit() forwards to `sides` (hittable.rs:362-365)
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref { list(b, &self.sides) }
}
// impl Hittable for Vec<Arc<HittableSS>> (hittable.rs:380):   fn flatten(&self, b) -> vk_ref { list(b, self) }
impl ConstantMedium { // hittable.rs:451
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let boundary = hittable(b, &self.boundary);
        let material = material(b, &self.phase_function);
        b.media.push(vk_medium { boundary, neg_inv_density: self.neg_inv_density, material });
        make_ref(VK_KIND_MEDIUM, b.media.len() - 1)
    }
}
impl Translate {     // hittable.rs:506
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let child = hittable(b, &self.ptr);
        b.translates.push(vk_translate { child, offset: v3(self.offset) });
        make_ref(VK_KIND_TRANSLATE, b.translates.len() - 1)
    }
}
fn rotate(b: &mut FlatBuilder, ptr: &Arc<HittableSS>, axis: u32, sin_theta: f32, cos_theta: f32) -> vk_ref {
    let child = hittable(b, ptr);
    b.rotates.push(vk_rotate { child, axis, sin_theta, cos_theta });
    make_ref(VK_KIND_ROTATE, b.rotates.len() - 1)
}
impl RotateX { fn flatten(&self, b: &mut FlatBuilder) -> vk_ref { rotate(b, &self.ptr, 0, self.sin_theta, self.cos_theta) } }  // hittable.rs:675
impl RotateY { fn flatten(&self, b: &mut FlatBuilder) -> vk_ref { rotate(b, &self.ptr, 1, self.sin_theta, self.cos_theta) } }  // hittable.rs:578
impl RotateZ { fn flatten(&self, b: &mut FlatBuilder) -> vk_ref { rotate(b, &self.ptr, 2, self.sin_theta, self.cos_theta) } }  // hittable.rs:764
// impl Hittable for Bowser (scene.rs:536-549):   fn flatten(&self, b) -> vk_ref { hittable(b, &self.parts) }

// ---- accel.rs -------------------------------------------------------------------------------------
impl BVHNode {       // accel.rs:58
    fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
        let idx = b.bvh.len();
        b.bvh.push(vk_bvh_node { bb_min: v3(self.bb.min), bb_max: v3(self.bb.max), left: 0, right: 0 });
        let left = hittable(b, &self.left);       // pre-order, left first: the order BVHNode::hit visits them (accel.rs:64-70)
        let right = hittable(b, &self.right);
        b.bvh[idx].left = left;
        b.bvh[idx].right = right;
        make_ref(VK_KIND_BVH, idx)
    }
}

// ---- material.rs ----------------------------------------------------------------------------------
fn push_material(b: &mut FlatBuilder, kind: u32, texture: u32, param: f32, a: u32, bb: u32) -> u32 {
    b.materials.push(vk_material { kind, texture, param, a, b: bb });
    (b.materials.len() - 1) as u32
}
impl Lambertian { fn flatten(&self, b: &mut FlatBuilder) -> u32 { let t = texture(b, &self.albedo); push_material(b, 0, t, 0.0, 0, 0) } }
impl Metal { fn flatten(&self, b: &mut FlatBuilder) -> u32 { let t = texture(b, &self.albedo); push_material(b, 1, t, self.fuzz, 0, 0) } }
impl Dielectric { fn flatten(&self, b: &mut FlatBuilder) -> u32 { push_material(b, 2, 0, self.ref_idx, 0, 0) } }
impl DiffuseLight { fn flatten(&self, b: &mut FlatBuilder) -> u32 { let t = texture(b, &self.emit); push_material(b, 3, t, 0.0, 0, 0) } }
impl Isotropic { fn flatten(&self, b: &mut FlatBuilder) -> u32 { let t = texture(b, &self.albedo); push_material(b, 4, t, 0.0, 0, 0) } }
impl SpecDiffuse {
    fn flatten(&self, b: &mut FlatBuilder) -> u32 {
        let (s, d) = (material(b, &self.specular), material(b, &self.diffuse));
        push_material(b, 5, 0, self.pct, s, d)
    }
}
fn push_texture(b: &mut FlatBuilder, kind: u32, color: [f32; 3], a: u32, bb: u32, scale: f32) -> u32 {
    b.textures.push(vk_texture { kind, color, a, b: bb, scale });
    (b.textures.len() - 1) as u32
}
impl SolidColor { fn flatten(&self, b: &mut FlatBuilder) -> u32 { push_texture(b, 0, v3(self.color_value), 0, 0, 0.0) } }
impl Checker {
    fn flatten(&self, b: &mut FlatBuilder) -> u32 {
        let (o, e) = (texture(b, &self.odd), texture(b, &self.even));
        push_texture(b, 1, [0.0; 3], o, e, 0.0)
    }
}
impl ImageTexture {  // the decoded RGB8 buffer crosses as it is (material.rs:261-279); it must outlive vk_scene_create
    fn flatten(&self, b: &mut FlatBuilder) -> u32 {
        b.images.push(vk_image { width: self.width as u32, height: self.height as u32, rgb: self.buf.as_ptr() });
        push_texture(b, 2, [0.0; 3], (b.images.len() - 1) as u32, 0, 0.0)
    }
}
impl NoiseTexture {
    fn flatten(&self, b: &mut FlatBuilder) -> u32 {
        let p = &self.noise;
        let mut rec = vk_perlin { ranvec: [[0.0; 3]; 256], perm_x: [0; 256], perm_y: [0; 256], perm_z: [0; 256] };
        for i in 0..256 {
            rec.ranvec[i] = v3(p.random_data[i]);
            rec.perm_x[i] = p.perm_x[i] as u32; rec.perm_y[i] = p.perm_y[i] as u32; rec.perm_z[i] = p.perm_z[i] as u32;
        }
        b.perlins.push(rec);
        push_texture(b, 3, [0.0; 3], (b.perlins.len() - 1) as u32, 0, self.scale)
    }
}
