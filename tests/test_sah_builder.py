"""the opt-in SAH builder of the host mirror (`<scene>+sah`).  Same objects and geometry as the
reference builder (BVHNode::new, accel.rs:98-136), another tree; closest hits do not depend on the tree's shape,
so for scenes without in-traversal draws (no ConstantMedium) the rendered image is the reference tree's image."""
import numpy as np

from vecchio_amd import HostScene, ffi


def leaves(desc, ref, out):
    kind = ref >> 28
    if kind == ffi.VK_KIND_BVH:
        n = desc.bvh[ref & ffi.VK_REF_INDEX_MASK]
        leaves(desc, n.left, out)
        leaves(desc, n.right, out)
    else:
        out.append(ref)


def test_same_objects_other_tree(built):
    a, b = HostScene("random_spheres_iow", 1), HostScene("random_spheres_iow+sah", 1)
    da, db = a.desc.contents, b.desc.contents
    assert da.n_spheres == db.n_spheres and da.n_materials == db.n_materials
    sa = np.array([(s.center[0], s.center[1], s.center[2], s.radius) for s in da.spheres[:da.n_spheres]])
    sb = np.array([(s.center[0], s.center[1], s.center[2], s.radius) for s in db.spheres[:db.n_spheres]])
    assert np.array_equal(np.sort(sa, axis=0), np.sort(sb, axis=0))          # identical geometry (same random stream)
    la, lb = [], []
    leaves(da, da.world, la)
    leaves(db, db.world, lb)
    assert len(set(la)) == da.n_spheres == len(set(lb))                      # every sphere is a leaf of both trees
    # node boxes enclose their children
    for d in (da, db):
        for i in range(d.n_bvh):
            n = d.bvh[i]
            for c in (n.left, n.right):
                if c >> 28 == ffi.VK_KIND_BVH:
                    m = d.bvh[c & ffi.VK_REF_INDEX_MASK]
                    assert all(n.bb_min[k] <= m.bb_min[k] and n.bb_max[k] >= m.bb_max[k] for k in range(3))


def test_image_is_shape_independent_and_cheaper(built, oracle):
    imgs, boxes = [], []
    for name in ("random_spheres_iow", "random_spheres_iow+sah"):
        hs = HostScene(name, 1)
        cam = hs.next_camera()
        p = hs.params(96, 2, 50, seed=5)
        img, cnt = oracle.render(hs.desc, cam, p, threads=8)
        imgs.append(img)
        boxes.append(cnt.as_dict()["n_aabb"])
    assert np.allclose(imgs[0], imgs[1], rtol=0, atol=1e-6)                  # same closest hits, same draws, same paths
    assert boxes[1] < 0.7 * boxes[0]                                         # and far fewer AxisBB::hit calls
