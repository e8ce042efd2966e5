// Stand-in for <boost/optional.hpp> (build check only): apps/m17-demod.cpp includes it and uses std::optional.
#pragma once
