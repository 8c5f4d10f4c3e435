/*
 * vit_hip_experiments.h -- entry points that exist only in MEASUREMENT builds of libvit_hip.so (never in the shipped library).
 *
 *   make -C viterbidecodercpp_amd/csrc EXTRA=-DVIT_HIP_CLOCK_STAMPS OUT=../libvit_hip_stamps.so BUILD=build_stamps
 *
 * No reference counterpart: the reference times its phases with std::chrono on the host (examples/run_benchmark.cpp:272-281).
 */
#ifndef VIT_HIP_EXPERIMENTS_H
#define VIT_HIP_EXPERIMENTS_H
#ifdef __cplusplus
extern "C" {
#endif
#ifdef VIT_HIP_CLOCK_STAMPS
/* Every register-plan update launched after this call stamps lane 0 of each wavefront into d_stamps, [tiles][6] uint64 on the
 * device: { s_memtime at entry, s_memrealtime at entry, s_memtime at exit, s_memrealtime at exit, XCC id, HW id }.  The ratio
 * of the two differences times the constant clock's rate (hipDeviceAttributeWallClockRate) is the shader clock the wave itself
 * ran at -- alone or beside a chainback kernel, per XCD (bench.py: clock_mhz.update_kernel).  NULL switches the stamps off. */
int vit_hip_experiment_clock_stamps(void* d_stamps);
#endif
#ifdef __cplusplus
}
#endif
#endif
