R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06b; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 500 python3 tools/ab_bench.py --postings 1e9 --unit-ints 16384 --rounds 4 --reps 3 base=$V/base.so lazy1=$V/lazy1.so > $OUT/ab_single_lazy.txt 2>&1
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 ff4=$V/ff4.so ffd=$V/ffd.so ff8=$V/ff8.so > $OUT/ab_multi_pack.txt 2>&1
cd /tmp && export TMPDIR=/tmp
AB_NO_ASSERT=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/ldspmc -- python3 $R/tools/ab_bench.py --postings 1e9 --unit-ints 16384 --rounds 1 --reps 2 base=$V/base.so lds16=$V/lds16.so lds24=$V/lds24.so lds26=$V/lds26.so lds30=$V/lds30.so > $OUT/ldspmc.log 2>&1
cd $R
python3 tools/lds_conflict_table.py $OUT/ldspmc 2 base lds16 lds24 lds26 lds30 > $OUT/lds_conflicts.txt 2>&1
AB_NO_ASSERT=1 timeout 500 python3 tools/ab_bench.py --postings 1e9 --unit-ints 16384 --rounds 3 --reps 3 base=$V/base.so lds16=$V/lds16.so lds24=$V/lds24.so lds26=$V/lds26.so lds30=$V/lds30.so > $OUT/ab_lds_times.txt 2>&1
rm -rf $OUT/ldspmc/*/*.db 2>/dev/null
find $OUT/ldspmc -size +3M -delete
tail -4 $OUT/ab_single_lazy.txt; tail -5 $OUT/ab_multi_pack.txt; cat $OUT/lds_conflicts.txt; tail -7 $OUT/ab_lds_times.txt
