// scene.cpp — the reference's scene builders (scene.rs), restated in C++ as the harness side
// of the boundary.  Geometry, materials, cameras and the ORDER of random draws follow
// scene.rs line for line; the draws come from the seeded build stream (vecchio_host.hpp).
#include "vecchio_host.hpp"

namespace vecchio {

static TextureP solid(Vec3 c) { return std::make_shared<SolidColor>(c); }
static MaterialP lambert(Vec3 c) { return std::make_shared<Lambertian>(solid(c)); }

// scene.rs:93-165
SceneConfig balls_demo() {
    SceneConfig cfg;
    auto &world = cfg.world;
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, 0.0f, -1.0f), 0.5f, lambert(Vec3(0.1f, 0.2f, 0.5f))));
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, -100.5f, -1.0f), 100.0f, lambert(Vec3(0.8f, 0.8f, 0.8f))));
    world.push_back(std::make_shared<Sphere>(Vec3(1.0f, 0.0f, -1.0f), 0.5f, std::make_shared<Metal>(solid(Vec3(0.8f, 0.6f, 0.2f)), 0.3f)));
    world.push_back(std::make_shared<Sphere>(Vec3(-1.0f, 0.0f, -1.0f), 0.5f, std::make_shared<Dielectric>(1.5f)));
    world.push_back(std::make_shared<Sphere>(Vec3(-1.0f, 0.0f, -1.0f), -0.45f, std::make_shared<Dielectric>(1.5f)));
    auto light_shape = Rect::XZRect(-6.0f, 6.0f, -6.0f, 6.0f, 8.0f, std::make_shared<DiffuseLight>(solid(Vec3::new_const(4.0f))));
    world.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    float aspect_ratio = 16.0f / 9.0f;
    cfg.cam_iter = FixedCamera(camera_new(Vec3(0.0f, 2.0f, 10.0f), Vec3(0.0f, 1.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio,
        0.0f, 10.0f, 0.0f, 1.0f));
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// scene.rs:167-284 (HEAD state)
SceneConfig random_spheres_demo() {
    SceneConfig cfg;
    auto &world = cfg.world;
    auto checker = std::make_shared<Checker>(solid(Vec3(0.1f, 0.1f, 0.1f)), solid(Vec3(0.9f, 0.9f, 0.9f)));
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, std::make_shared<Lambertian>(checker)));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose_mat = gen_f32();
            float cx = (float)a + 0.9f * gen_f32();
            float cz = (float)b + 0.9f * gen_f32();
            Vec3 center(cx, 0.2f, cz);
            if ((center - Vec3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
                if (choose_mat < 0.8f) {
                    Vec3 r1 = Vec3::random();
                    Vec3 r2 = Vec3::random();
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, lambert(r1 * r2)));
                } else if (choose_mat < 0.95f) {
                    auto albedo = solid(Vec3::random_range(0.5f, 1.0f));
                    float fuzz = gen_range(0.0f, 0.5f);
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, std::make_shared<Metal>(albedo, fuzz)));
                } else {
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, std::make_shared<Dielectric>(1.5f)));
                }
            }
        }
    }
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, 1.0f, 0.0f), 1.0f, std::make_shared<Dielectric>(1.5f)));
    world.push_back(std::make_shared<Sphere>(Vec3(-4.0f, 1.0f, 0.0f), 1.0f,
                                             std::make_shared<Lambertian>(ImageTexture::open("assets/earthmap.png"))));
    world.push_back(std::make_shared<Sphere>(Vec3(4.0f, 1.0f, 0.0f), 1.0f, std::make_shared<Metal>(solid(Vec3(0.7f, 0.6f, 0.5f)), 0.0f)));
    auto light = std::make_shared<DiffuseLight>(solid(Vec3(1.0f, 0.77f, 0.56f) * 2.0f));
    auto light_shape = Rect::XZRect(-11.0f, 11.0f, -11.0f, 11.0f, 8.0f, light);
    world.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    float aspect_ratio = 16.0f / 9.0f;
    cfg.cam_iter = RotatingCamera(Vec3(0.0f, 1.5f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f, aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f,
                                  2.5f, 25.0f, 20.0f, 0.5f, 360.0f);
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// The InOneWeekend-tag scene (BASELINE C1/C2; C5 with a larger grid): random_spheres_demo's
// layout (scene.rs:184-240) minus its HEAD additions (checker ground 171-175, earth texture
// 229-231, sky light 244-251, rotating camera 254-281), with the book's IOW camera, grey
// ground and brown diffuse big sphere.  Sphere-only, Material::scatter integrator, sky.
SceneConfig random_spheres_iow(int grid_half) {
    SceneConfig cfg;
    auto &world = cfg.world;
    bool big = grid_half > 11;
    float ground_r = big ? 100000.0f : 1000.0f;
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, -ground_r, 0.0f), ground_r, lambert(Vec3(0.5f, 0.5f, 0.5f))));
    for (int a = -grid_half; a < grid_half; a++) {
        for (int b = -grid_half; b < grid_half; b++) {
            float choose_mat = gen_f32();
            float cx = (float)a + 0.9f * gen_f32();
            float cz = (float)b + 0.9f * gen_f32();
            Vec3 center(cx, 0.2f, cz);
            if ((center - Vec3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
                if (choose_mat < 0.8f) {
                    Vec3 r1 = Vec3::random();
                    Vec3 r2 = Vec3::random();
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, lambert(r1 * r2)));
                } else if (choose_mat < 0.95f) {
                    auto albedo = solid(Vec3::random_range(0.5f, 1.0f));
                    float fuzz = gen_range(0.0f, 0.5f);
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, std::make_shared<Metal>(albedo, fuzz)));
                } else {
                    world.push_back(std::make_shared<Sphere>(center, 0.2f, std::make_shared<Dielectric>(1.5f)));
                }
            }
        }
    }
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, 1.0f, 0.0f), 1.0f, std::make_shared<Dielectric>(1.5f)));
    world.push_back(std::make_shared<Sphere>(Vec3(-4.0f, 1.0f, 0.0f), 1.0f, lambert(Vec3(0.4f, 0.2f, 0.1f))));
    world.push_back(std::make_shared<Sphere>(Vec3(4.0f, 1.0f, 0.0f), 1.0f, std::make_shared<Metal>(solid(Vec3(0.7f, 0.6f, 0.5f)), 0.0f)));
    float aspect_ratio = 16.0f / 9.0f;
    if (!big) {
        cfg.cam_iter = FixedCamera(camera_new(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f, aspect_ratio,
            0.1f, 10.0f, 0.0f, 1.0f));
    } else {
        // stress scene: square image, camera pulled back and raised so the field fills the frame
        aspect_ratio = 1.0f;
        float k = (float)grid_half / 11.0f;
        cfg.cam_iter = FixedCamera(camera_new(Vec3(13.0f * k * 0.35f, 2.0f * k * 1.2f, 3.0f * k * 0.35f), Vec3(0.0f, 0.0f, 0.0f),
            Vec3(0.0f, 1.0f, 0.0f),
                                              40.0f, aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f));
    }
    cfg.aspect_ratio = aspect_ratio;
    cfg.integrator = VK_INTEGRATOR_SCATTER;
    cfg.background = VK_BACKGROUND_SKY;
    return cfg;
}

// scene.rs:286-338
SceneConfig perlin_demo() {
    SceneConfig cfg;
    auto &world = cfg.world;
    auto pertext = std::make_shared<Lambertian>(std::make_shared<NoiseTexture>(2.0f));
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, pertext));
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, 2.0f, 0.0f), 2.0f, pertext));
    auto light_shape = Rect::XZRect(-6.0f, 6.0f, -6.0f, 6.0f, 8.0f, std::make_shared<DiffuseLight>(solid(Vec3::new_const(4.0f))));
    world.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    float aspect_ratio = 16.0f / 9.0f;
    cfg.cam_iter = FixedCamera(camera_new(Vec3(0.0f, 2.0f, 10.0f), Vec3(0.0f, 1.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio,
        0.0f, 10.0f, 0.0f, 1.0f));
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// scene.rs:340-549: struct Bowser — 5 image-textured rects, a grey bottom, 26 Boxys, in a BVHNode.
// (The reference wraps the BVH in a Hittable that forwards hit/bounding_box: a no-op here.)
// assets/bowser_*.png and assets/twitter.png: decoded copies, see ImageTexture::open.
static HittableP bowser_new(float x, float y, float z) {
    std::vector<HittableP> world;
    auto img = [](const char *path) { return std::make_shared<Lambertian>(ImageTexture::open(path)); };
    const float y0 = y - 1.875f, zf = z + 4.5f;
    world.push_back(Rect::XYRect(x - 2.0f, x + 2.0f, y0 + 1.0f, y0 + 4.0f, zf - 3.0f, img("assets/bowser_face.png")));     // face
    world.push_back(Rect::XZRect(x - 2.0f, x + 2.0f, zf - 6.0f, zf - 3.0f, y0 + 4.0f, img("assets/bowser_top.png")));     // top
    world.push_back(Rect::XYRect(x - 2.0f, x + 2.0f, y0 + 1.0f, y0 + 4.0f, zf - 6.0f, img("assets/bowser_back.png")));     // back
    auto side = img("assets/bowser_side.png");
    world.push_back(Rect::YZRect(y0 + 1.0f, y0 + 4.0f, zf - 6.0f, zf - 3.0f, x - 2.0f, side));       // sides share one texture
    world.push_back(Rect::YZRect(y0 + 1.0f, y0 + 4.0f, zf - 6.0f, zf - 3.0f, x + 2.0f, side));
    auto grey = lambert(Vec3(0.278f, 0.387f, 0.438f));
    world.push_back(Rect::XZRect(x - 2.0f, x + 2.0f, zf - 6.0f, zf - 3.0f, y0 + 1.0f, grey));        // bottom
    auto box = [&](float ax, float ay, float az, float bx, float by, float bz, MaterialP m) {
        world.push_back(std::make_shared<Boxy>(Vec3(x + ax, y0 + ay, zf + az), Vec3(x + bx, y0 + by, zf + bz), m));
    };
    // feet
    box(-1.5f, 0.5f, -4.75f, -0.5f, 1.0f, -4.25f, grey);
    box(0.5f, 0.5f, -4.75f, 1.5f, 1.0f, -4.25f, grey);
    box(-1.5f, 0.25f, -4.75f, -0.5f, 0.5f, -3.5f, grey);
    box(0.5f, 0.25f, -4.75f, 1.5f, 0.5f, -3.5f, grey);
    // arms
    auto brown = lambert(Vec3(0.4f, 0.2f, 0.1f));
    box(-2.25f, 1.75f, -4.65f, -2.00f, 2.75f, -4.35f, brown);
    box(-2.50f, 1.75f, -4.65f, -2.25f, 2.50f, -4.35f, brown);
    box(-2.75f, 1.75f, -4.65f, -2.50f, 2.25f, -4.35f, brown);
    box(2.00f, 1.75f, -4.65f, 2.25f, 2.75f, -4.35f, brown);
    box(2.25f, 1.75f, -4.65f, 2.50f, 2.50f, -4.35f, brown);
    box(2.50f, 1.75f, -4.65f, 2.75f, 2.25f, -4.35f, brown);
    // face rim
    auto lightgrey = lambert(Vec3(0.601f, 0.687f, 0.723f));
    box(-2.0f, 3.875f, -3.00f, 2.0f, 4.00f, -2.875f, lightgrey);
    box(-2.0f, 1.0f, -3.00f, 2.0f, 1.125f, -2.875f, lightgrey);
    box(-2.0f, 1.125f, -3.00f, -1.875f, 3.875f, -2.875f, lightgrey);
    box(1.875f, 1.125f, -3.00f, 2.0f, 3.875f, -2.875f, lightgrey);
    // ports on the back
    box(-1.875f, 1.625f, -6.125f, -0.875f, 1.75f, -6.0f, lightgrey);
    box(-1.875f, 1.125f, -6.125f, -0.875f, 1.25f, -6.0f, lightgrey);
    box(-1.875f, 1.25f, -6.125f, -1.750f, 1.625f, -6.0f, lightgrey);
    box(-1.0f, 1.25f, -6.125f, -0.875f, 1.625f, -6.0f, lightgrey);
    box(0.875f, 1.625f, -6.125f, 1.875f, 1.75f, -6.0f, lightgrey);
    box(0.875f, 1.125f, -6.125f, 1.875f, 1.25f, -6.0f, lightgrey);
    box(1.750f, 1.25f, -6.125f, 1.875f, 1.625f, -6.0f, lightgrey);
    box(0.875f, 1.25f, -6.125f, 1.0f, 1.625f, -6.0f, lightgrey);
    return BVHNode::build(world);
}

// scene.rs:551-628
SceneConfig bowser_demo() {
    SceneConfig cfg;
    auto &world = cfg.world;
    auto checker = std::make_shared<Checker>(solid(Vec3(0.1f, 0.1f, 0.1f)), solid(Vec3(0.9f, 0.9f, 0.9f)));
    world.push_back(std::make_shared<Sphere>(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, std::make_shared<Lambertian>(checker)));
    // Translate(RotateX(RotateY(RotateZ(Bowser, 0), 0), 0), (0, 1.625, -4.5)): three zero-angle rotations, kept
    world.push_back(std::make_shared<Translate>(RotateX(RotateY(RotateZ(bowser_new(0.0f, 0.0f, 0.0f), 0.0f), 0.0f), 0.0f),
                                                Vec3(0.0f, 1.625f, -4.5f)));
    auto light_shape = Rect::XYRect(-2.0f, 2.0f, 1.0f, 4.0f, 3.0f,
                                    // image-textured emitter, scene.rs:585-590
                                    std::make_shared<DiffuseLight>(ImageTexture::open("assets/twitter.png")));
    world.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    float aspect_ratio = 16.0f / 9.0f;
    cfg.cam_iter = RotatingCamera(Vec3(0.0f, 2.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f, aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f,
                                  2.5f, -35.0f, 20.0f, 0.5f, 360.0f - 35.0f);
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// scene.rs:630-730
SceneConfig cornell_box() {
    SceneConfig cfg;
    auto &world = cfg.world;
    auto red = lambert(Vec3(0.65f, 0.05f, 0.05f));
    auto white = lambert(Vec3(0.73f, 0.73f, 0.73f));
    auto green = lambert(Vec3(0.12f, 0.45f, 0.15f));
    auto light = std::make_shared<DiffuseLight>(solid(Vec3(15.0f, 15.0f, 15.0f)));
    world.push_back(std::make_shared<FlipFace>(Rect::YZRect(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, green)));
    world.push_back(Rect::YZRect(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, red));
    world.push_back(std::make_shared<FlipFace>(Rect::XZRect(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, white)));
    world.push_back(Rect::XZRect(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white));
    world.push_back(std::make_shared<FlipFace>(Rect::XYRect(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white)));
    auto box1 = std::make_shared<Boxy>(Vec3::new_const(0.0f), Vec3(165.0f, 330.0f, 165.0f), white);
    world.push_back(std::make_shared<Translate>(RotateY(box1, 15.0f), Vec3(265.0f, 0.0f, 295.0f)));
    world.push_back(std::make_shared<Sphere>(Vec3(190.0f, 90.0f, 190.0f), 90.0f, std::make_shared<Dielectric>(1.5f)));
    auto light_shape = Rect::XZRect(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, light);
    world.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    float aspect_ratio = 1.0f;
    cfg.cam_iter = FixedCamera(camera_new(Vec3(278.0f, 278.0f, -800.0f), Vec3(278.0f, 278.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 40.0f,
        aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f));
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// scene.rs:732-874
SceneConfig final_scene() {
    SceneConfig cfg;
    auto &objects = cfg.world;
    std::vector<HittableP> boxes1;
    auto ground = lambert(Vec3(0.48f, 0.83f, 0.53f));
    const int BOXES_PER_SIDE = 20;
    for (int i = 0; i < BOXES_PER_SIDE; i++) {
        for (int j = 0; j < BOXES_PER_SIDE; j++) {
            float w = 100.0f;
            float x0 = -1000.0f + (float)i * w;
            float z0 = -1000.0f + (float)j * w;
            float y0 = 0.0f;
            float x1 = x0 + w;
            float z1 = z0 + w;
            float y1 = gen_range(1.0f, 101.0f);
            boxes1.push_back(std::make_shared<Boxy>(Vec3(x0, y0, z0), Vec3(x1, y1, z1), ground));
        }
    }
    objects.push_back(BVHNode::build(boxes1));
    auto light = std::make_shared<DiffuseLight>(solid(Vec3::new_const(7.0f)));
    auto light_shape = Rect::XZRect(123.0f, 423.0f, 147.0f, 412.0f, 554.0f, light);
    objects.push_back(std::make_shared<FlipFace>(light_shape));
    cfg.lights.push_back(light_shape);
    Vec3 center1(400.0f, 400.0f, 200.0f);
    Vec3 center2 = center1 + Vec3(30.0f, 0.0f, 0.0f);
    objects.push_back(std::make_shared<MovingSphere>(center1, center2, 0.0f, 1.0f, 50.0f, lambert(Vec3(0.7f, 0.3f, 0.1f))));
    objects.push_back(std::make_shared<Sphere>(Vec3(260.0f, 150.0f, 45.0f), 50.0f, std::make_shared<Dielectric>(1.5f)));
    objects.push_back(std::make_shared<Sphere>(Vec3(0.0f, 150.0f, 145.0f), 50.0f,
        std::make_shared<Metal>(solid(Vec3(0.8f, 0.8f, 0.9f)), 10.0f)));
    auto boundary1 = std::make_shared<Sphere>(Vec3(360.0f, 150.0f, 145.0f), 70.0f, std::make_shared<Dielectric>(1.5f));
    objects.push_back(boundary1);
    objects.push_back(std::make_shared<ConstantMedium>(boundary1, 0.2f, solid(Vec3(0.2f, 0.4f, 0.9f))));
    auto boundary2 = std::make_shared<Sphere>(Vec3::new_const(0.0f), 5000.0f, std::make_shared<Dielectric>(1.5f));
    objects.push_back(std::make_shared<ConstantMedium>(boundary2, 0.0001f, solid(Vec3::new_const(1.0f))));
    auto emat = std::make_shared<Lambertian>(ImageTexture::open("assets/earthmap.png"));  // scene.rs:821-823
    objects.push_back(std::make_shared<Sphere>(Vec3(400.0f, 200.0f, 400.0f), 100.0f, emat));
    auto pertext = std::make_shared<NoiseTexture>(0.1f);
    objects.push_back(std::make_shared<Sphere>(Vec3(220.0f, 280.0f, 300.0f), 80.0f, std::make_shared<Lambertian>(pertext)));
    std::vector<HittableP> boxes2;
    auto white = lambert(Vec3::new_const(0.73f));
    for (int i = 0; i < 1000; i++) boxes2.push_back(std::make_shared<Sphere>(Vec3::random_range(0.0f, 165.0f), 10.0f, white));
    objects.push_back(std::make_shared<Translate>(RotateY(BVHNode::build(boxes2), 15.0f), Vec3(-100.0f, 270.0f, 395.0f)));
    float aspect_ratio = 1.0f;
    cfg.cam_iter = FixedCamera(camera_new(Vec3(478.0f, 278.0f, -600.0f), Vec3(278.0f, 278.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 40.0f,
        aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f));
    cfg.aspect_ratio = aspect_ratio;
    return cfg;
}

// The configuration sample/thenextweek.png was rendered with (README.md:9-15): the same objects, but the TheNextWeek tag
// predates the PDF integrator — ray_color = emitted + attenuation * ray_color(scattered) through Material::scatter
// (material.rs:21-28), black background.  (The tag itself is unreadable; the picture's light quad, fog and sphere
// placement are HEAD's, which tests/test_golden_nextweek.py checks.)
SceneConfig final_scene_nextweek() {
    SceneConfig cfg = final_scene();
    cfg.integrator = VK_INTEGRATOR_SCATTER;
    cfg.background = VK_BACKGROUND_SOLID;
    cfg.background_color = Vec3::new_const(0.0f);
    return cfg;
}

}  // namespace vecchio
