// ORACLE (test infrastructure). Selects the field back end of this translation unit:
//   default           Goldilocks / GoldilocksExt2  (gl.hpp, namespace orc,   C symbols orc_*)    the goldilocks test family
//   -DORC_FIELD_BN254 bn256::Fr with E = F = Fr    (fr.hpp, namespace orcbn, C symbols orcbn_*)  the bn254 test family
//                                                  [REF bfv-gkr/src/sk_encryption_circuit.rs:539-540, 614-626]
// Everything above this header (transcript, polys, sum-check, Lasso, GKR, BFV circuit) is written once against F / E.
#pragma once
#if defined(ORC_FIELD_BN254)
#include "fr.hpp"
#else
#include "gl.hpp"
#endif
