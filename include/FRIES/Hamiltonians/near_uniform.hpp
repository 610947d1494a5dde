/*! \file  FRIES/Hamiltonians/near_uniform.hpp for the MI355X build.  frisys_mol.cpp includes this header without calling into it; the
 * near-uniform excitation sampling itself (doub_multin / sing_multin, near_uniform.cpp) runs inside the engine's FCIQMC path
 * (fries_fciqmc_setup with distribution NU, fries_amd/csrc/fciqmc.hip) and has no host-callable form in this build. */
#ifndef near_uniform_h
#define near_uniform_h
#include <FRIES/Hamiltonians/molecule.hpp>
#endif /* near_uniform_h */
