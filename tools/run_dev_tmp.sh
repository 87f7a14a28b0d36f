bash tools/pmc_clock.sh real build_variants/dev/libapgpu.so
APGPU_DEBUG_STRIDE0=1 bash tools/pmc_clock.sh stride0 build_variants/dev/libapgpu.so
APGPU_DEBUG_MAXITERS=1 bash tools/pmc_clock.sh mi1 build_variants/dev/libapgpu.so
