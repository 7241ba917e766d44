//! Raw bindings to `librt_amd.so` (declarations of `include/rt_amd.h`, one to one).
//!
//! Every function returns `RT_OK` or a negative `RT_ERR_*` and never unwinds; `rt_last_error`
//! gives the thread-local message of the last failure. Pointers are host pointers unless the
//! parameter name says `device`.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_NO_DEVICE: c_int = -2;
pub const RT_ERR_HIP: c_int = -3;
pub const RT_ERR_IO: c_int = -4;
pub const RT_ERR_OOM: c_int = -5;

pub const RT_MAT_OPAQUE: i32 = 0;
pub const RT_MAT_DIELECTRIC: i32 = 1;
pub const RT_OBJ_MESH: i32 = 0;
pub const RT_OBJ_SPHERE: i32 = 1;

pub const RT_FLAG_NO_JITTER: u32 = 1;
pub const RT_FLAG_TIMING: u32 = 2;
pub const RT_FLAG_COUNT_WORK: u32 = 4;
pub const RT_FLAG_NO_REGEN: u32 = 8;
pub const RT_FLAG_MEGAKERNEL: u32 = 16;
pub const RT_FLAG_PIPELINE: u32 = 32;
pub const RT_FLAG_STORE: u32 = 64;

pub const RT_TREE_HOST: u32 = 0;
pub const RT_TREE_NONE: u32 = 1;

pub const RT_BUILD_LBVH: u32 = 0;
pub const RT_BUILD_PLOC: u32 = 1;
pub const RT_BUILD_DEFAULT: u32 = RT_BUILD_PLOC;

pub const RT_GROUP_MAX: u32 = 16;
pub const RT_NOALLOC_SCRATCH_MIB: u32 = 320;
pub const RT_SPLIT_SAMPLES: u32 = 0;
pub const RT_SPLIT_TILES: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_material {
    pub kind: i32,
    pub color_tex: i32,
    pub normal_tex: i32,
    pub reserved: i32,
    pub metallic: f64,
    pub diffuse: f64,
    pub how_metallic: f64,
    pub emission: [f64; 3],
    pub color: [f64; 3],
    pub ior: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_texture {
    pub rgba: *const u8,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_scene_desc {
    pub n_objects: u32,
    pub object_kind: *const i32,
    pub materials: *const rt_material,
    pub n_triangles: u64,
    pub tri_vertices: *const f64,
    pub tri_normals: *const f64,
    pub tri_uvs: *const f64,
    pub tri_obj: *const i32,
    pub tri_idx: *const i32,
    pub n_spheres: u32,
    pub spheres: *const f64,
    pub sphere_obj: *const i32,
    pub n_textures: u32,
    pub textures: *const rt_texture,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct rt_hit {
    pub t: f64,
    pub u: f64,
    pub v: f64,
    pub obj: i32,
    pub tri: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub first_sample: u32,
    pub seed: u64,
    pub max_depth: i32,
    pub flags: u32,
    pub fov_y_deg: f64,
    pub dof_strength: f64,
    pub spp_total: u32,
    pub tile_size: u32,
    pub tile_rank: u32,
    pub tile_ranks: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_scene_info {
    pub n_triangles: u64,
    pub n_spheres: u32,
    pub n_objects: u32,
    pub n_textures: u32,
    pub bvh_nodes: u32,
    pub bvh_leaves: u32,
    pub bvh_depth: u32,
    pub uploaded: u32,
    pub bvh_spheres: u32,
    pub reserved: u32,
    pub device_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_stats {
    pub rays: u64,
    pub node_visits: u64,
    pub tri_tests: u64,
    pub sphere_tests: u64,
    pub shaded_hits: u64,
    pub paths: u64,
    pub waves: u64,
    pub wave_iters_inner: u64,
    pub wave_iters_leaf: u64,
    pub wave_iters_reject: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_timing {
    pub ms_paths: f32,
    pub ms_resolve: f32,
    pub path_launches: u32,
    pub resolve_launches: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_launch_info {
    pub kind: u32,
    pub items: u32,
    pub ms: f32,
    pub folded: u32,
    pub rays: u64,
    pub node_visits: u64,
    pub tri_tests: u64,
    pub waves: u64,
    pub wave_iters_inner: u64,
    pub wave_iters_leaf: u64,
    pub wave_iters_reject: u64,
}

pub enum rt_image {}
pub enum rt_mesh {}
pub enum rt_objects {}
pub enum rt_scene {}
pub enum rt_group {}

extern "C" {
    pub fn rt_mesh_load_obj(path: *const c_char, out: *mut *mut rt_mesh) -> c_int;
    pub fn rt_mesh_clone(mesh: *const rt_mesh, out: *mut *mut rt_mesh) -> c_int;
    pub fn rt_mesh_destroy(mesh: *mut rt_mesh);
    pub fn rt_mesh_counts(mesh: *const rt_mesh, n_vertices: *mut u64, n_triangles: *mut u64, n_uvs: *mut u64) -> c_int;
    pub fn rt_mesh_rotate_x(mesh: *mut rt_mesh, angle: f64) -> c_int;
    pub fn rt_mesh_rotate_y(mesh: *mut rt_mesh, angle: f64) -> c_int;
    pub fn rt_mesh_rotate_z(mesh: *mut rt_mesh, angle: f64) -> c_int;
    pub fn rt_mesh_translate(mesh: *mut rt_mesh, t: *const f64) -> c_int;

    pub fn rt_image_load_png(path: *const c_char, out: *mut *mut rt_image) -> c_int;
    pub fn rt_image_destroy(img: *mut rt_image);
    pub fn rt_image_info(img: *const rt_image, width: *mut u32, height: *mut u32, rgba: *mut *const u8) -> c_int;
    pub fn rt_image_write_png(path: *const c_char, rgba: *const u8, width: u32, height: u32) -> c_int;

    pub fn rt_objects_create(out: *mut *mut rt_objects) -> c_int;
    pub fn rt_objects_destroy(objs: *mut rt_objects);
    pub fn rt_objects_add_texture(objs: *mut rt_objects, rgba: *const u8, width: u32, height: u32) -> c_int;
    pub fn rt_objects_push_mesh(objs: *mut rt_objects, mesh: *const rt_mesh, mat: *const rt_material) -> c_int;
    pub fn rt_objects_push_sphere(objs: *mut rt_objects, centre: *const f64, radius: f64, mat: *const rt_material) -> c_int;

    pub fn rt_scene_file_load(path: *const c_char, out: *mut *mut rt_objects, params: *mut rt_render_params, its: *mut u32) -> c_int;
    pub fn rt_render_params_default(p: *mut rt_render_params);

    pub fn rt_scene_from_objects(objs: *const rt_objects, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_create(desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_from_objects_with(objs: *const rt_objects, tree: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_create_with(desc: *const rt_scene_desc, tree: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_get_info(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_export_flat(scene: *const rt_scene, tri_vertices: *mut f64, tri_normals: *mut f64, tri_uvs: *mut f64, tri_obj: *mut i32, tri_idx: *mut i32) -> c_int;
    pub fn rt_scene_export_objects(scene: *const rt_scene, object_kind: *mut i32, materials: *mut rt_material, spheres: *mut f64, sphere_obj: *mut i32) -> c_int;
    pub fn rt_scene_export_texture(scene: *const rt_scene, id: u32, width: *mut u32, height: *mut u32, rgba: *mut u8) -> c_int;
    pub fn rt_scene_export_bvh(scene: *const rt_scene, node_words: *mut u32, leaf_order: *mut i32) -> c_int;
    pub fn rt_scene_export_grid_bvh(scene: *const rt_scene, node_words: *mut u32, lo_step: *mut f32, usable: *mut i32) -> c_int;

    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_upload(scene: *mut rt_scene, device: c_int) -> c_int;
    pub fn rt_scene_rebuild_on_device(scene: *mut rt_scene, device: c_int) -> c_int;
    pub fn rt_scene_rebuild_on_device_with(scene: *mut rt_scene, device: c_int, method: u32) -> c_int;
    pub fn rt_trace(scene: *mut rt_scene, org3: *const f64, dir3: *const f64, n: usize, out: *mut rt_hit) -> c_int;
    pub fn rt_primary_hits(scene: *mut rt_scene, p: *const rt_render_params, obj: *mut i32, tri: *mut i32, t: *mut f64, u: *mut f64, v: *mut f64) -> c_int;
    pub fn rt_render_accumulate(scene: *mut rt_scene, p: *const rt_render_params, canvas: *mut f64, rays_out: *mut u64) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, p: *const rt_render_params, device_canvas: *mut f64, stream: *mut c_void) -> c_int;
    pub fn rt_render_device_noalloc(scene: *mut rt_scene, p: *const rt_render_params, device_canvas: *mut f64, stream: *mut c_void) -> c_int;
    pub fn rt_render_sum_f32_device(scene: *mut rt_scene, p: *const rt_render_params, device_sum: *mut f32, stream: *mut c_void) -> c_int;
    pub fn rt_scene_wait(scene: *mut rt_scene) -> c_int;
    pub fn rt_scene_stream_wait(scene: *mut rt_scene, stream: *mut c_void) -> c_int;
    pub fn rt_group_create(scene: *const rt_scene, devices: *const c_int, n_devices: u32, out: *mut *mut rt_group) -> c_int;
    pub fn rt_group_destroy(group: *mut rt_group);
    pub fn rt_group_size(group: *const rt_group, n_members: *mut u32) -> c_int;
    pub fn rt_group_render_accumulate(group: *mut rt_group, p: *const rt_render_params, split: u32, canvas: *mut f64, rays_out: *mut u64) -> c_int;
    pub fn rt_group_render_device(group: *mut rt_group, p: *const rt_render_params, split: u32, device_canvas: *mut f64, stream: *mut c_void) -> c_int;
    pub fn rt_group_wait(group: *mut rt_group, rays_out: *mut u64) -> c_int;
    pub fn rt_group_member_info(group: *const rt_group, member: u32, device: *mut c_int, rays: *mut u64, ms: *mut f64) -> c_int;
    pub fn rt_scene_last_stats(scene: *mut rt_scene, out: *mut rt_stats) -> c_int;
    pub fn rt_scene_last_timing(scene: *mut rt_scene, out: *mut rt_timing) -> c_int;
    pub fn rt_scene_last_launches(scene: *mut rt_scene, out: *mut rt_launch_info, capacity: u32, n_out: *mut u32) -> c_int;
    pub fn rt_quantize_rgba8(canvas: *const f64, width: u32, height: u32, its: u32, out: *mut u8) -> c_int;
    pub fn rt_selftest_division(n: u64, seed: u64, mismatches: *mut u64) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_source_digest() -> *const c_char;
}
