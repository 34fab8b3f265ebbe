// `verifier` = the interactive verifier bound to the HIP-backed prover (reference src/verifier.hpp:11)
#pragma once
#include "prover.hpp"
#include "verifier.hpp"
typedef verifierT<prover> verifier;
