/* sgrl_render.h -- C ABI of the off-screen renderer behind BatchedModularVecEnv.get_images() (gfx950).
 *
 * What it replaces: `SubprocVecEnv.get_images()` (reference src/subproc_vec_env.py:70-73), i.e. every worker's
 * `env.render(mode='rgb_array')` = MuJoCo's OpenGL off-screen renderer through gym 0.17.2 / mujoco-py (third-party, absent here).
 * This is a ray caster over the same scene description (ground plane, spheres, capsules at the bodies' world poses) with the
 * reference's camera set-up (`viewer_setup`, <env>.py:166-170: tracked body, distance, look-at height, elevation -20 deg).
 * PIXEL PARITY WITH MuJoCo's RASTERISER IS NEITHER CLAIMED NOR PINNED: it is a visualisation aid held by geometric property
 * tests (tests/test_render_gpu.py).
 *
 * One image = one camera + a list of geoms in world coordinates.  All pointers are DEVICE pointers owned by the caller.
 */
#ifndef SGRL_RENDER_H
#define SGRL_RENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* geom record, 16 floats: type (0 plane z = 0 with a checker pattern, 2 sphere, 3 capsule) | centre xyz | unit axis xyz (capsule)
 * | radius | half length | rgb | 3 unused */
#define SGRL_RENDER_GEOM_FLOATS 16
/* camera record, 13 floats: eye xyz | forward xyz | right xyz | up xyz | tan(fovy / 2) */
#define SGRL_RENDER_CAM_FLOATS 13

/* rgb[n_img][height][width][3] (uint8).  geoms[n_img][max_geoms][16], n_geoms[n_img], cams[n_img][13]. */
int sgrl_render(const float* geoms, const int32_t* n_geoms, int max_geoms, const float* cams, int n_img, int width, int height,
                uint8_t* rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif
