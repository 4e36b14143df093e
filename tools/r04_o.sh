#!/bin/bash
# round 4, run O: Fenton fast, FIB_FENTON_FEWER bits 2 (the outward current's chain negated in its constants: a literal instead of a
# scalar register in its last multiply-add) and 3 (the excitation test on the difference a = U - u_c instead of on U): 3 = shipped so far.
# (Both forms were built from a scratch copy of csrc/ and did not gain: they are NOT in the tree — profiles/r04_fewer_instructions.txt has the numbers.)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3 4; do
  for b in issue_fenton_fw3 issue_fenton_fw7 issue_fenton_fw11 issue_fenton_fw15; do timeout -k 5 60 ./$b 20000 252 || echo "$b FAILED rc $?"; done
  for b in mt_ab_fw3 mt_ab_fw7 mt_ab_fw11 mt_ab_fw15; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done
done > ../../gpurun_out/r04/o_fenton_fewer.txt 2>&1
cd ../..
grep issue_ gpurun_out/r04/o_fenton_fewer.txt | sed 's/:.*steps: /: /' | cut -c1-60 | sort | uniq -c | sort -k2 | head -20
grep mt_ab gpurun_out/r04/o_fenton_fewer.txt | sort -s -k1,1 | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
