LIBS="base=simka_amd/lib/libsimka_base.so new=" SAMPLES=6 bash scripts/pmc_ab.sh
PMC="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" LIBS="base=simka_amd/lib/libsimka_base.so new=" SAMPLES=6 bash scripts/pmc_ab.sh
for v in basepp newpp; do echo "== $v"; SIMKA_LIB_OVERRIDE=$PWD/simka_amd/lib/libsimka_$v.so timeout 300 python bench.py --lanes 1 --no-two-streams --workload c3 --samples 4 --steps 1 --warmup 1 --no-cpu-baseline --no-from-host --no-e2e 2>&1 | grep "k_skm_count_fast" | tail -4; done
