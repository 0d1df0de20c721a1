V=signed-heat-3d_amd/lib/variants
for rep in 1 2 3; do python3 tools/r04_ab.py bunny_small.obj:4:64 "default=" "setup0=SHM_SETUP_PRIO=0" "p1=SHM_GRID_LIB=$V/libshm_grid_p1.so" "p1s0=SHM_GRID_LIB=$V/libshm_grid_p1.so;SHM_SETUP_PRIO=0" "p2s0=SHM_GRID_LIB=$V/libshm_grid_p2.so;SHM_SETUP_PRIO=0" "p2=SHM_GRID_LIB=$V/libshm_grid_p2.so" "r04=SHM_GRID_LIB=$V/libshm_grid_r04.so"; done
