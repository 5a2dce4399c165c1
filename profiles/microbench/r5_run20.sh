ROUNDS=2 STEPS=12 bash profiles/microbench/ab_lib.sh abl/lib_scold.so abl/lib_scnew.so
bash profiles/microbench/r5_full_tests.sh
bash profiles/microbench/r5_run21.sh
