// flatten.rs — the ONE method the shim adds to each of vecchio's traits, because they offer no
// introspection (hittable.rs:33-42, material.rs:20-41,228-230) and the concrete types' fields are
// module-private: each impl lives next to its type (hittable.rs / material.rs / accel.rs).
//
// UNCOMPILED SOURCE (no Rust toolchain in the build image).  The compiled, tested twin is
// vecchio_amd/host/host.cpp (`X::flatten(FlatBuilder&)`), line for line the same logic.
//
// trait Hittable  { ...; fn flatten(&self, b: &mut FlatBuilder) -> vk_ref; }
// trait Material  { ...; fn flatten(&self, b: &mut FlatBuilder) -> u32; }
// trait Texture   { ...; fn flatten(&self, b: &mut FlatBuilder) -> u32; }
use crate::ffi::*;

// Arc identity -> record, so Arc::clone'd objects flatten once
fn key<T: ?Sized>(a: &std::sync::Arc<T>) -> usize { std::sync::Arc::as_ptr(a) as *const u8 as usize }
pub fn hittable(b: &mut FlatBuilder, h: &std::sync::Arc<crate::hittable::HittableSS>) -> vk_ref {
    if let Some(r) = b.seen.get(&key(h)) { return *r; }
    let r = h.flatten(b);
    b.seen.insert(key(h), r);
    r
}
// material(b, &Arc<MaterialSS>) / texture(b, &Arc<TextureSS>) are the same three lines.

// ---- hittable.rs -----------------------------------------------------------------------------
// impl Hittable for Sphere:
//     fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
//         b.spheres.push(vk_sphere { center: [self.center.x, self.center.y, self.center.z], radius: self.radius,
//                                    material: material(b, &self.material) });
//         make_ref(VK_KIND_SPHERE, b.spheres.len() - 1)
//     }
// impl Hittable for MovingSphere: push vk_moving_sphere { center0, center1, time0, time1, radius, material } -> VK_KIND_MOVING_SPHERE
// impl Hittable for Rect:         push vk_rect { c0, c1, d0, d1, k, axis0, axis1, axis2, material } -> VK_KIND_RECT
// impl Hittable for FlipFace:     hittable(b, &self.ptr) ^ VK_REF_FLIP            // only negates `front` (hittable.rs:299-308)
// impl Hittable for Boxy:         flatten the six `sides` into b.list_items, push vk_list { first, count } -> VK_KIND_LIST
// impl Hittable for Vec<Arc<HittableSS>>: same as Boxy
// impl Hittable for ConstantMedium: push vk_medium { boundary: hittable(b,&self.boundary), neg_inv_density, material(b,&self.phase_function) }
// impl Hittable for Translate:    push vk_translate { child: hittable(b,&self.ptr), offset } -> VK_KIND_TRANSLATE
// impl Hittable for RotateX/Y/Z:  push vk_rotate { child, axis: 0/1/2, sin_theta, cos_theta } -> VK_KIND_ROTATE
// impl Hittable for Bowser (scene.rs:340-549): hittable(b, &self.parts)
//
// ---- accel.rs --------------------------------------------------------------------------------
// impl Hittable for BVHNode:
//     fn flatten(&self, b: &mut FlatBuilder) -> vk_ref {
//         let idx = b.bvh.len();
//         b.bvh.push(vk_bvh_node { bb_min: [self.bb.min.x, ..], bb_max: [self.bb.max.x, ..], left: 0, right: 0 });
//         let l = hittable(b, &self.left);
//         let r = hittable(b, &self.right);
//         b.bvh[idx].left = l; b.bvh[idx].right = r;
//         make_ref(VK_KIND_BVH, idx)
//     }
//
// ---- material.rs -----------------------------------------------------------------------------
// Lambertian   -> vk_material { kind: 0, texture: texture(b,&self.albedo), param: 0.0, a: 0, b: 0 }
// Metal        -> kind 1, texture, param = fuzz
// Dielectric   -> kind 2, param = ref_idx
// DiffuseLight -> kind 3, texture = emit
// Isotropic    -> kind 4, texture = albedo
// SpecDiffuse  -> kind 5, param = pct, a = material(b,&self.specular), b = material(b,&self.diffuse)
// SolidColor   -> vk_texture { kind: 0, color }
// Checker      -> kind 1, a = texture(b,&self.odd), b = texture(b,&self.even)
// ImageTexture -> kind 2, a = index of vk_image { width, height, rgb: self.buf.as_ptr() }
// NoiseTexture -> kind 3, a = index of vk_perlin { ranvec, perm_x, perm_y, perm_z }, scale
