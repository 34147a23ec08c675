// TEST DOUBLE (see third_party_decls.hpp)
#pragma once
#include "../../third_party_decls.hpp"
