//! The reference's main() (src/main.rs:1054-1128) with its two hot calls replaced:
//! `sphere_scene()` + `Scene::from_objects` (1077, 750) become a scene handle, and
//! `render::<WIDTH, HEIGHT>(&mut buffer, &scene)` (1099) becomes `rt_render_accumulate`.
//! The window is left out; the display step (1106-1112) is `rt_quantize_rgba8`.
use rt_amd_sys::*;
use std::ffi::{CStr, CString};
use std::ptr;

const WIDTH: u32 = 1600;
const HEIGHT: u32 = 1200;
const NUM_SAMPLES_PER_PIXEL: u32 = 10;

fn check(rc: i32, what: &str) {
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
        panic!("{} failed ({}): {}", what, rc, msg);
    }
}

/// sphere_scene(), src/main.rs:1270-1450, through the library's mesh and object calls.
unsafe fn sphere_scene() -> *mut rt_scene {
    let mut objs = ptr::null_mut();
    check(rt_objects_create(&mut objs), "rt_objects_create");

    let mut img = ptr::null_mut();
    let tex_path = CString::new("spot_texture.png").unwrap();
    check(rt_image_load_png(tex_path.as_ptr(), &mut img), "rt_image_load_png");
    let (mut w, mut h, mut px) = (0u32, 0u32, ptr::null());
    check(rt_image_info(img, &mut w, &mut h, &mut px), "rt_image_info");
    let spot_text = rt_objects_add_texture(objs, px, w, h);
    check(spot_text, "rt_objects_add_texture");
    rt_image_destroy(img);

    let opaque = |color: [f64; 3], color_tex: i32, metallic: f64, how_metallic: f64, diffuse: f64| rt_material {
        kind: RT_MAT_OPAQUE, color_tex, normal_tex: -1, reserved: 0, metallic, diffuse, how_metallic,
        emission: [0.0; 3], color, ior: 1.0,
    };
    let glass = |ior: f64| rt_material { kind: RT_MAT_DIELECTRIC, color_tex: -1, normal_tex: -1, ior, ..Default::default() };

    let mut floor = ptr::null_mut();
    let floor_path = CString::new("triangle_floor.obj").unwrap();
    check(rt_mesh_load_obj(floor_path.as_ptr(), &mut floor), "rt_mesh_load_obj");
    check(rt_objects_push_mesh(objs, floor, &opaque([0.7, 0.7, 0.7], -1, 0.1, 0.0, 1.0)), "rt_objects_push_mesh");
    rt_mesh_destroy(floor);

    let mut mirror = ptr::null_mut();
    let spot_path = CString::new("spot_triangulated.obj").unwrap();
    check(rt_mesh_load_obj(spot_path.as_ptr(), &mut mirror), "rt_mesh_load_obj");
    check(rt_mesh_rotate_y(mirror, 220.0 * std::f64::consts::PI / 180.0), "rt_mesh_rotate_y");
    check(rt_mesh_translate(mirror, [1.0, 0.0, -2.5].as_ptr()), "rt_mesh_translate");
    check(rt_objects_push_mesh(objs, mirror, &opaque([0.0; 3], spot_text, 0.0, 0.0, 1.0)), "rt_objects_push_mesh");
    rt_mesh_destroy(mirror);

    let spheres: [([f64; 3], f64, rt_material); 6] = [
        ([-0.3, -0.2, -2.8], 0.5, opaque([0.7, 0.7, 0.75], -1, 0.0, 1.0, 0.9)),
        ([-1.0, -0.5, -2.8], 0.2, opaque([1.0, 1.0, 1.0], -1, 0.0, 1.0, 0.9)),
        ([-1.5, -0.4, -2.3], 0.3, opaque([0.0, 0.0, 0.6], -1, 0.0, 1.0, 0.9)),
        ([0.3, -0.4, -1.4], 0.3, glass(1.5)),
        ([-0.15, -0.6, -2.0], 0.1, opaque([0.7, 0.0, 0.0], -1, 0.0, 0.0, 1.0)),
        ([0.0, 0.3, -1.9], 0.1, glass(1.5)),
    ];
    for (c, r, m) in spheres.iter() {
        check(rt_objects_push_sphere(objs, c.as_ptr(), *r, m), "rt_objects_push_sphere");
    }

    let mut scene = ptr::null_mut();
    check(rt_scene_from_objects(objs, &mut scene), "rt_scene_from_objects");
    rt_objects_destroy(objs);
    check(rt_scene_upload(scene, 0), "rt_scene_upload");
    scene
}

fn main() {
    unsafe {
        let scene = sphere_scene();
        // src/main.rs:1079, as one contiguous buffer: buffer[(v * WIDTH + u) * 3 + c]
        let mut buffer = vec![0.0f64; (WIDTH * HEIGHT * 3) as usize];
        let mut rgba = vec![0u8; (WIDTH * HEIGHT * 4) as usize];
        let mut p = rt_render_params::default();
        rt_render_params_default(&mut p);
        p.width = WIDTH;
        p.height = HEIGHT;
        p.spp = NUM_SAMPLES_PER_PIXEL;

        let mut its = 0u32;
        while its < 16 {
            p.first_sample = its * NUM_SAMPLES_PER_PIXEL; // continue each pixel's sample stream
            check(rt_render_accumulate(scene, &p, buffer.as_mut_ptr(), ptr::null_mut()), "rt_render_accumulate");
            its += 1;
            check(rt_quantize_rgba8(buffer.as_ptr(), WIDTH, HEIGHT, its, rgba.as_mut_ptr()), "rt_quantize_rgba8");
        }
        let out = CString::new("render.png").unwrap();
        check(rt_image_write_png(out.as_ptr(), rgba.as_ptr(), WIDTH, HEIGHT), "rt_image_write_png");
        rt_scene_destroy(scene);
    }
}
