// oracle/stdaln_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
// Restatement of the one stdaln entry the PacBio code uses outside the MSA: aln_stdaln(s1, s2, &aln_param_pacbio, ALN_TYPE_GLOBAL, 1)
// (Thirdparty/stdaln.c:231-248 parameters, :364-546 aln_global_core, :780-862 aln_stdaln_aux), called by
// SAIPBSelfCorrectTree::mergeTwoSeedsUsingHash (PacBio/SAIPBSelfCTree.cpp:186-194) to pick among several merged sequences.
// Pinned to the reference's object code: tests/test_oracle_vs_ref.py::test_stdaln_global_* and tests/golden/stdaln_kats.json.
#pragma once
#include <string>

namespace lrsc_oracle {

struct StdalnGlobal {
    int matches;    // number of '|' in AlnAln::outm: aligned pairs of equal nucleotides
    int score;      // AlnAln::score
    int path_len;   // AlnAln::path_len
};

// Banded global alignment (band 50 + the length difference), affine gaps (open 1, extend 1, end gaps 0), PacBio matrix
// (match 1, mismatch -8, anything against N -2).  Both strings non-empty.
StdalnGlobal stdaln_global_pacbio(const std::string& s1, const std::string& s2);

} // namespace lrsc_oracle
