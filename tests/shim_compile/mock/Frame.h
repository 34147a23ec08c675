// TEST DOUBLE (see reference_decls.hpp)
#pragma once
#include "reference_decls.hpp"
