# hip_compiler.jl -- Julia host side of the MI355X evaluator back end.
#
# Drop into src/backend/ of FeynmanDiagram.jl and `include("hip_compiler.jl")`
# from src/backend/compiler.jl after the existing includes (compiler.jl:16-18).
# It adds
#
#     Compilers.compile_hip(graphs; root=[id(g) for g in graphs], backend=:isa)
#         -> (f::GraphFunc, leafmap::Dict{Int,G})
#
# with the *same* `leafmap` (index of `leafVal` -> leaf graph object) that
# `Compilers.compile` / `to_julia_str` return (static.jl:98-133), so
# `FrontEnds.leafstates(leaf_maps, ...)` (frontends.jl:115-232) works unchanged.
#
#     f(root::AbstractVector, leafVal::AbstractVector)   one sample; mutates root,
#                                                        returns the last root written
#     f(root::AbstractMatrix, leafVal::AbstractMatrix)   B samples: leafVal is B x L,
#                                                        root is B x R (column-major
#                                                        Julia matrices = the "leaf
#                                                        major" layout of the C ABI)
#
# STATUS: UNVERIFIED.  Julia is not installed in the build environment or on the
# GPU box, so this file has never been executed.  The C ABI it binds
# (include/fdg.h) is exercised by the Python/ctypes twin (capi.py) in tests/.
#
# The traversal below restates to_julia_str's loop over the arrays of the node
# table instead of over text; it must stay in lock step with static.jl:98-133.

const _libfdg = get(ENV, "FDG_LIB", "libfdg.so")

struct _FdgGraphDesc
    n_leaf::UInt32
    n_node::UInt32
    n_root::UInt32
    n_edge::UInt32
    op::Ptr{UInt8}
    power::Ptr{Int32}
    child_off::Ptr{UInt32}
    child_idx::Ptr{UInt32}
    child_fac::Ptr{Float64}
    root_slot::Ptr{UInt32}
end

const FDG_NO_ROOT = 0xffffffff
const FDG_SPEC_ISA = Cuint(4)
const FDG_SPEC_AUTOTUNE = Cuint(8)

_fdg_check(rc) = rc == 0 ? nothing :
    error("fdg error $rc: " * unsafe_string(ccall((:fdg_last_error, _libfdg), Cstring, ())))

mutable struct GraphFunc
    handle::Ptr{Cvoid}
    n_leaf::Int
    n_root::Int
    last_root::Int          # 1-based index of the root written last, 0 if none
    cache_dir::Union{Nothing,String}   # the JIT cache directory given to compile_hip (nothing: the library's per-user default); later specialisations use it too
    function GraphFunc(h, L, R, last, cache_dir=nothing)
        f = new(h, L, R, last, cache_dir)
        finalizer(x -> ccall((:fdg_graph_destroy, _libfdg), Cint, (Ptr{Cvoid},), x.handle), f)
        return f
    end
end

_opcode(::Type{ComputationalGraphs.Sum}) = (0x00, Int32(0))
_opcode(::Type{ComputationalGraphs.Prod}) = (0x01, Int32(0))
_opcode(::Type{ComputationalGraphs.Power{N}}) where {N} = (0x02, Int32(N))
_opcode(op::Type) = error(                       # same failure as static.jl:6-11
    "Static representation for computational graph nodes with operator $(op) not yet implemented! ")

"""
    lower_to_table(graphs; root) -> (arrays..., leafmap)

Statement order, leaf numbering and root mapping of `to_julia_str` (static.jl:98-133).
"""
function lower_to_table(graphs::AbstractVector{G}; root::AbstractVector{Int}=[id(g) for g in graphs]) where {G<:AbstractGraph}
    leaf_index = Dict{Int,Int}()      # id -> 0-based leaf index
    node_index = Dict{Int,Int}()      # id -> 0-based internal index
    leafmap = Dict{Int,G}()
    order = G[]
    stmt_pos = Dict{Int,Int}()        # id -> position of its statement in the text to_julia_str would emit
    for graph in graphs
        for g in PostOrderDFS(graph)
            g_id = id(g)
            if isempty(subgraphs(g))
                haskey(leaf_index, g_id) && continue
                leaf_index[g_id] = length(leaf_index)
                leafmap[length(leaf_index)] = g
            else
                haskey(node_index, g_id) && continue
                _opcode(operator(g))
                node_index[g_id] = length(order)
                push!(order, g)
            end
            stmt_pos[g_id] = length(stmt_pos)
        end
    end
    L = length(leaf_index)
    vidx(gid) = haskey(leaf_index, gid) ? leaf_index[gid] : L + node_index[gid]
    op = UInt8[]; power = Int32[]; off = UInt32[0]; idx = UInt32[]; fac = Float64[]
    for g in order
        o, p = _opcode(operator(g))
        push!(op, o); push!(power, p)
        for (sg, f) in zip(subgraphs(g), subgraph_factors(g))
            push!(idx, UInt32(vidx(id(sg)))); push!(fac, Float64(f))
        end
        push!(off, UInt32(length(idx)))
    end
    root_slot = fill(UInt32(FDG_NO_ROOT), length(root))
    last_root, last_rank = 0, -1
    for (k, rid) in enumerate(root)
        findfirst(==(rid), root) == k || continue             # findfirst (static.jl:112)
        (haskey(leaf_index, rid) || haskey(node_index, rid)) || continue
        root_slot[k] = UInt32(vidx(rid))
        rank = stmt_pos[rid]          # `root[k] = g` follows g's own statement (static.jl:126-128): the last one is the call's value
        if rank > last_rank
            last_rank, last_root = rank, k
        end
    end
    return L, op, power, off, idx, fac, root_slot, leafmap, last_root
end

"""
    compile_hip(graphs; root, backend=:isa, autotune=false, groups=nothing, cache_dir=nothing, association=:static)

`association = :eval` makes the handle reproduce the interpreter `eval!` (src/computational_graph/eval.jl:1-3,15-39: every operand is
scaled by its factor before it enters the fold, `Prod = (w1*f1) * (w2*f2) * ...`) instead of the generated code of `Compilers.compile`
(src/backend/static.jl:13-46) -- `fdg_graph_set_association(h, FDG_ASSOC_INTERP)`; the two differ only where a `Prod` has a factor other
than +-1 on its second or a later operand.

`groups` (optional `Dict{Int,Int}`: node id => tag) is the scheduling hint of
`fdg_graph_set_schedule_groups`: pass the coefficient -> original-node map of `taylorexpansion!`
(`to_coeff_map`, src/utility.jl:105-135) so that all Taylor coefficients of one node are evaluated
together.  It never changes a value.
"""
function compile_hip(graphs::AbstractVector{<:AbstractGraph};
    root::AbstractVector{Int}=[id(g) for g in graphs], backend::Symbol=:isa, autotune::Bool=false,
    groups::Union{Nothing,Dict{Int,Int}}=nothing, cache_dir::Union{Nothing,String}=nothing, association::Symbol=:static)
    association in (:static, :eval) || error("association must be :static or :eval")
    L, op, power, off, idx, fac, root_slot, leafmap, last_root = lower_to_table(graphs; root=root)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve op power off idx fac root_slot begin
        desc = Ref(_FdgGraphDesc(UInt32(L), UInt32(length(op)), UInt32(length(root_slot)), UInt32(length(idx)),
            pointer(op), pointer(power), pointer(off), pointer(idx), pointer(fac), pointer(root_slot)))
        _fdg_check(ccall((:fdg_graph_create, _libfdg), Cint, (Ref{_FdgGraphDesc}, Ref{Ptr{Cvoid}}), desc, h))
    end
    if association == :eval      # FDG_ASSOC_INTERP = 1; before any specialisation
        _fdg_check(ccall((:fdg_graph_set_association, _libfdg), Cint, (Ptr{Cvoid}, Cint), h[], Cint(1)))
    end
    if !isnothing(groups)
        # internal nodes in statement order, exactly the order lower_to_table numbered them
        seen = Set{Int}(); tags = Dict{Int,UInt32}(); grp = UInt32[]
        for graph in graphs, g in PostOrderDFS(graph)
            (isempty(subgraphs(g)) || id(g) in seen) && continue
            push!(seen, id(g))
            key = get(groups, id(g), -id(g))
            push!(grp, get!(tags, key, UInt32(length(tags))))
        end
        _fdg_check(ccall((:fdg_graph_set_schedule_groups, _libfdg), Cint, (Ptr{Cvoid}, Ptr{UInt32}, UInt32), h[], grp, UInt32(length(grp))))
    end
    if backend != :interp
        flags = backend == :isa ? (FDG_SPEC_ISA | (autotune ? FDG_SPEC_AUTOTUNE : Cuint(0))) : Cuint(0)
        cdir = isnothing(cache_dir) ? C_NULL : cache_dir
        rc = ccall((:fdg_graph_specialize, _libfdg), Cint, (Ptr{Cvoid}, Cstring, Cuint), h[], cdir, flags)
        if rc == -2 && backend == :isa      # FDG_E_UNSUPPORTED (the ISA back end covers every operator of the reference; kept for future ones)
            rc = ccall((:fdg_graph_specialize, _libfdg), Cint, (Ptr{Cvoid}, Cstring, Cuint), h[], cdir, Cuint(0))
        elseif rc == 0 && backend == :isa && L > 1 && length(op) <= 4000
            # a handle without a row-major variant of the ISA kernel (fdg_kernel_info.has_rm == 0: fewer than 16 leaves, the
            # tiny-graph configuration): HIP-source companion for row-major [B, L] input (FDG_SPEC_ROW_MAJOR_COMPANION = 16)
            ki = zeros(UInt8, 256)              # sizeof(fdg_kernel_info) = 168 (48 + 3*8 + 4*3*4 + 4*4, then the pool / rl fields), rounded up
            _fdg_check(ccall((:fdg_graph_kernel_info, _libfdg), Cint, (Ptr{Cvoid}, Ptr{UInt8}), h[], ki))
            has_rm = reinterpret(UInt32, ki[125:128])[1]      # offset 124: has_acc at 120, has_rm at 124
            if has_rm == 0
                rc = ccall((:fdg_graph_specialize, _libfdg), Cint, (Ptr{Cvoid}, Cstring, Cuint), h[], cdir, Cuint(16))
            end
        end
        _fdg_check(rc)
    end
    return GraphFunc(h[], L, length(root_slot), last_root, cache_dir), leafmap
end

# one sample: the calling convention of the generated eval_graph!(root, leafVal)
function (f::GraphFunc)(root::AbstractVector{Float64}, leafVal::AbstractVector{Float64})
    length(leafVal) >= f.n_leaf || throw(BoundsError(leafVal, f.n_leaf))
    length(root) >= f.n_root || throw(BoundsError(root, f.n_root))
    lv = Vector{Float64}(leafVal[1:f.n_leaf]); rt = Vector{Float64}(root[1:f.n_root])
    _fdg_check(ccall((:fdg_eval, _libfdg), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), f.handle, lv, rt, 1))
    root[1:f.n_root] .= rt
    return f.last_root == 0 ? nothing : root[f.last_root]
end

# B samples, host matrices (B x L in, B x R out, column-major): H2D, eval, D2H
function (f::GraphFunc)(root::Matrix{Float64}, leafVal::Matrix{Float64})
    B = size(leafVal, 1)
    size(leafVal, 2) >= f.n_leaf || throw(BoundsError(leafVal, (1, f.n_leaf)))
    size(root) == (B, f.n_root) || throw(DimensionMismatch("root must be B x R"))
    # column-major matrices as they are: sample stride 1, value stride B (no transposition copy on either side;
    # the device sees them leaf-major, the evaluator's fast layout)
    _fdg_check(ccall((:fdg_eval_strided, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64),
        f.handle, leafVal, 1, B, root, 1, B, B))
    return root
end

# B samples, device pointers (e.g. from AMDGPU.jl ROCArrays of size B x L / B x R):
# strides in elements; a column-major B x L matrix is (sample stride 1, leaf stride B)
function eval_device!(f::GraphFunc, d_root::Ptr{Float64}, d_leaf::Ptr{Float64}, B::Integer;
    leaf_strides=(1, B), root_strides=(1, B), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_eval_device, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64, Ptr{Cvoid}),
        f.handle, d_leaf, leaf_strides[1], leaf_strides[2], d_root, root_strides[1], root_strides[2], B, stream))
end

# Element types other than Float64 (the function Compilers.compile returns is generic in eltype(leafVal)): the per-type kernel
# is compiled on first use; a ComplexF64 element is the pair (re, im), strides count elements
const _FDG_DT = Dict{DataType,Cint}(Float64 => 0, Float32 => 1, ComplexF64 => 2, ComplexF32 => 3)
function eval_device!(f::GraphFunc, d_root::Ptr{T}, d_leaf::Ptr{T}, B::Integer;
    leaf_strides=(1, B), root_strides=(1, B), stream::Ptr{Cvoid}=C_NULL) where {T<:Union{Float32,ComplexF64,ComplexF32}}
    dt = _FDG_DT[T]
    # (flag 4 = FDG_SPEC_ISA: ComplexF64 batches whose rows are contiguous -- leaf_strides = (L, 1) -- additionally get the graph spelled out on
    #  real and imaginary parts through the assembly back end; a column-major B x L Julia matrix takes the per-type kernel)
    cdir = isnothing(f.cache_dir) ? C_NULL : f.cache_dir
    _fdg_check(ccall((:fdg_graph_specialize_typed, _libfdg), Cint, (Ptr{Cvoid}, Cint, Cstring, Cuint), f.handle, dt, cdir, Cuint(4)))
    _fdg_check(ccall((:fdg_eval_device_typed, _libfdg), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}),
        f.handle, dt, d_leaf, leaf_strides[1], leaf_strides[2], d_root, root_strides[1], root_strides[2], B, stream))
end

# Tile-major batches (include/fdg.h: fdg_eval_device_tiled): the leaves as an Array{Float64,3}(undef, 64, L, cld(B, 64)) -- sample in tile,
# leaf, tile -- and the roots as (64, R, cld(B, 64)); what a Monte-Carlo driver that owns its sample batch should allocate: one
# contiguous block of 512 L bytes per wave instead of 64 samples of each of L columns.  Strides in elements: (sample, value, tile).
function eval_device_tiled!(f::GraphFunc, d_root::Ptr{Float64}, d_leaf::Ptr{Float64}, B::Integer;
    leaf_strides=(1, 64, 64 * f.n_leaf), root_strides=(1, 64, 64 * f.n_root), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_eval_device_tiled, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64, Int64, Ptr{Cvoid}),
        f.handle, d_leaf, leaf_strides[1], leaf_strides[2], leaf_strides[3], d_root, root_strides[1], root_strides[2], root_strides[3], B, stream))
end
function accumulate_device_tiled!(f::GraphFunc, d_acc::Ptr{Float64}, d_leaf::Ptr{Float64}, d_weight::Ptr{Float64}, B::Integer;
    leaf_strides=(1, 64, 64 * f.n_leaf), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_accumulate_device_tiled, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Cvoid}),
        f.handle, d_leaf, leaf_strides[1], leaf_strides[2], leaf_strides[3], d_weight, d_acc, B, stream))
end
# device memory for a batch, backed by physical chunks of `chunk_bytes` (0: one allocation): fdg_batch_alloc / fdg_batch_free
function batch_alloc(bytes::Integer; chunk_bytes::Integer=0)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    _fdg_check(ccall((:fdg_batch_alloc, _libfdg), Cint, (Csize_t, Csize_t, Ref{Ptr{Cvoid}}), bytes, chunk_bytes, p))
    return p[]
end
batch_free(p::Ptr{Cvoid}) = _fdg_check(ccall((:fdg_batch_free, _libfdg), Cint, (Ptr{Cvoid},), p))

# The two arrays of a tile-major batch of `f` -- Array{Float64,3}(64, L, T) and (64, R, T) on the device -- with the root chunks chosen by
# timing f's own kernel on (leaf window, root chunk) pairs (fdg_batch_alloc_pair, include/fdg.h).  Returns (d_leaf, d_root, info bytes);
# release each pointer with batch_free.  What a Monte-Carlo driver that evaluates (not only accumulates) should allocate its batch with.
# layout = :leaf_major: a Julia Matrix pair B' x L / B' x R (B' = 64 * the info's chunk_tiles), for batches of up to a few tens of GB; :row_major: compile_Python's.
function batch_alloc_pair(f::GraphFunc, n_sample::Integer; chunk_bytes::Integer=0, calibrate::Bool=true, layout::Symbol=:tile_major)
    dl = Ref{Ptr{Cvoid}}(C_NULL); dr = Ref{Ptr{Cvoid}}(C_NULL)
    info = zeros(UInt8, 128)                 # fdg_batch_pair_info (120 bytes; the last two UInt32: level_reached -- 3 the full search, 2 a span cut
                                             # short by the free memory, 1 mapped in draw order, 0 not calibrated -- and span_gb)
    _fdg_check(ccall((:fdg_batch_alloc_pair, _libfdg), Cint, (Ptr{Cvoid}, Int64, Csize_t, Cuint, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}, Ptr{UInt8}),
        f.handle, n_sample, chunk_bytes, Cuint((calibrate ? 1 : 0) | (layout == :row_major ? 8 : 0) | (layout == :leaf_major ? 16 : 0)), dl, dr, info))
    return Ptr{Float64}(dl[]), Ptr{Float64}(dr[]), info
end

# tile_major!(d_tiled, d_src, B, C; strides): a device matrix in one of the reference's layouts -- a Julia column-major B x C Matrix{Float64}
# (strides = (1, B), the default) or compile_Python's row-major [B, C] (strides = (C, 1)) -- into the tile-major Array{Float64,3}(64, C, cld(B, 64))
# that eval_device_tiled! takes; from_tile_major! is the way back (roots).  One pass at copy speed each (fdg_repack_tile_major / fdg_unpack_tile_major):
# worth it for a batch that is evaluated several times; a producer that can write tile-major itself (the fused Monte-Carlo step does) should.
function tile_major!(d_tiled::Ptr{Float64}, d_src::Ptr{Float64}, B::Integer, C::Integer; strides=(1, B), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_repack_tile_major, _libfdg), Cint, (Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, UInt32, Ptr{Cvoid}),
        d_src, strides[1], strides[2], d_tiled, B, UInt32(C), stream))
end
function from_tile_major!(d_dst::Ptr{Float64}, d_tiled::Ptr{Float64}, B::Integer, C::Integer; strides=(1, B), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_unpack_tile_major, _libfdg), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, UInt32, Ptr{Cvoid}),
        d_tiled, d_dst, strides[1], strides[2], B, UInt32(C), stream))
end

# Options of a handle (what used to be FDG_* environment switches; the library reads the environment once per process): set_option!(f, "FDG_ISA_NO_POOL", "1")
set_option!(f::GraphFunc, name::AbstractString, value::AbstractString) =
    _fdg_check(ccall((:fdg_graph_set_option, _libfdg), Cint, (Ptr{Cvoid}, Cstring, Cstring), f.handle, name, value))
unset_option!(f::GraphFunc, name::AbstractString) =
    _fdg_check(ccall((:fdg_graph_set_option, _libfdg), Cint, (Ptr{Cvoid}, Cstring, Ptr{UInt8}), f.handle, name, C_NULL))

# acc[k] += sum_b weight[b] * root_k(b), everything on device
function accumulate_device!(f::GraphFunc, d_acc::Ptr{Float64}, d_leaf::Ptr{Float64}, d_weight::Ptr{Float64}, B::Integer;
    leaf_strides=(1, B), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_accumulate_device, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Cvoid}),
        f.handle, d_leaf, leaf_strides[1], leaf_strides[2], d_weight, d_acc, B, stream))
end

export compile_hip, GraphFunc, eval_device!, accumulate_device!, eval_device_tiled!, accumulate_device_tiled!, batch_alloc, batch_free, tile_major!, from_tile_major!

# ---- multi-GPU: one Julia process per GPU, ONE reduction of the accumulated observable ------------ #
# (include/fdg.h, "multi-GPU").  Rank 0 calls `comm_unique_id()` and ships the 128 bytes to the other
# ranks (MPI.jl `MPI.bcast`, a shared file, ...); every rank then builds its communicator with its own
# device current and, after its share of `accumulate_device!` calls, reduces `d_acc` in place.
const FDG_COMM_ID_BYTES = 128

function comm_unique_id()
    id = Vector{UInt8}(undef, FDG_COMM_ID_BYTES)
    _fdg_check(ccall((:fdg_comm_unique_id, _libfdg), Cint, (Ptr{UInt8}, Csize_t), id, FDG_COMM_ID_BYTES))
    return id
end

mutable struct Comm
    handle::Ptr{Cvoid}
    function Comm(id::Vector{UInt8}, rank::Integer, world::Integer)
        length(id) == FDG_COMM_ID_BYTES || error("unique id must be $FDG_COMM_ID_BYTES bytes")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        _fdg_check(ccall((:fdg_comm_create, _libfdg), Cint, (Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}), id, rank, world, h))
        c = new(h[])
        finalizer(x -> ccall((:fdg_comm_destroy, _libfdg), Cint, (Ptr{Cvoid},), x.handle), c)
        return c
    end
end

"""`d_acc[1:n]` <- sum over ranks (all ranks get it; `root >= 0`: only that rank), on `stream`."""
function reduce_device!(c::Comm, d_acc::Ptr{Float64}, n::Integer; root::Integer=-1, stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_reduce_device, _libfdg), Cint, (Ptr{Cvoid}, Ptr{Float64}, UInt32, Cint, Ptr{Cvoid}),
                     c.handle, d_acc, UInt32(n), Cint(root), stream))
    return nothing
end

# ---- fused Monte-Carlo step: leaves from (K, T) in registers, then the graph (include/fdg.h) --------- #
struct _FdgLeafTables
    n_leaf::UInt32
    n_basis::UInt32
    n_loop::UInt32
    dim::UInt32
    n_tau::UInt32
    leaf_type::Ptr{Int32}
    leaf_order::Ptr{Int32}
    tau_in::Ptr{Int32}
    tau_out::Ptr{Int32}
    loop_index::Ptr{Int32}
    basis::Ptr{Float64}
    kF::Float64
    beta::Float64
    lambda::Float64
end

"""
    specialize_fused!(f, leafType, leafOrder, leafInTau, leafOutTau, leafLoopIndex, loopbasis; dim=3, n_tau)

`leaf*` are one partition of `FrontEnds.leafstates` (1-based indices, `leafOrder` the order of the leaf's own kind),
`loopbasis` its deduplicated basis (`n_loop × n_basis`, columns as returned).  After this,
`mc_accumulate_device!(f, d_K, d_T, d_weight, d_acc, B; kF, beta, lambda)` runs leaves + graph + weighted sum in one kernel.
On a handle compiled with the optimizing back end (`compile(...; backend=:isa)`) that kernel is the back end's own: the
leaves are values of its program, computed in registers from the momenta and times.  It reads `d_K` (`B × n_loop*dim`)
and `d_T` (`B × n_tau`) in place when they are Julia column-major matrices -- the default `k_strides = t_strides = (1, B)`.
`kF`, `beta`, `lambda` are arguments of that kernel: one code object, assembled here, serves every parameter set.
"""
function specialize_fused!(f::GraphFunc, leafType::Vector{Int}, leafOrder::Vector{Int}, leafInTau::Vector{Int}, leafOutTau::Vector{Int},
    leafLoopIndex::Vector{Int}, loopbasis::Matrix{Float64}; dim::Int=3, n_tau::Int, cache_dir::Union{Nothing,String}=nothing,   # nothing: the library's per-user default (include/fdg.h)
    kF::Float64=0.0, beta::Float64=0.0, lambda::Float64=0.0)
    a = [Int32.(v) for v in (leafType, leafOrder, leafInTau, leafOutTau, leafLoopIndex)]
    bs = Matrix{Float64}(loopbasis)      # column-major n_loop × n_basis == row-major [n_basis][n_loop]
    GC.@preserve a bs begin
        tab = _FdgLeafTables(length(a[1]), size(bs, 2), size(bs, 1), dim, n_tau, pointer(a[1]), pointer(a[2]), pointer(a[3]),
            pointer(a[4]), pointer(a[5]), pointer(bs), kF, beta, lambda)
        _fdg_check(ccall((:fdg_graph_specialize_fused, _libfdg), Cint, (Ptr{Cvoid}, Ref{_FdgLeafTables}, Cstring, Cuint), f.handle, tab,
            isnothing(cache_dir) ? C_NULL : cache_dir, Cuint(0)))
    end
    return f
end

function mc_accumulate_device!(f::GraphFunc, d_K::Ptr{Float64}, d_T::Ptr{Float64}, d_weight::Ptr{Float64}, d_acc::Ptr{Float64}, B::Integer;
    kF::Float64, beta::Float64, lambda::Float64, k_strides=(1, B), t_strides=(1, B), stream::Ptr{Cvoid}=C_NULL)
    _fdg_check(ccall((:fdg_mc_accumulate_device, _libfdg), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Cvoid}),
        f.handle, d_K, k_strides[1], k_strides[2], d_T, t_strides[1], t_strides[2], kF, beta, lambda, d_weight, d_acc, B, stream))
    return nothing
end

"""
    leaf_eval_device!(d_leaf, leafType, leafOrder, leafInTau, leafOutTau, leafLoopIndex, loopbasis, d_K, d_T, B; dim, n_tau, kF, beta, lambda)

The leaf loop of `example/benchmark.jl:58-81` on the device (`fdg_leaf_eval_device`): fills the `B × L`
column-major leaf matrix at `d_leaf` from the loop momenta `d_K` (`B × (n_loop*dim)`, column-major) and times `d_T`
(`B × n_tau`).  Fermionic leaves of derivative order 0..5, interaction leaves of any order; type-0 leaves untouched.
"""
function leaf_eval_device!(d_leaf::Ptr{Float64}, leafType::Vector{Int}, leafOrder::Vector{Int}, leafInTau::Vector{Int}, leafOutTau::Vector{Int},
    leafLoopIndex::Vector{Int}, loopbasis::Matrix{Float64}, d_K::Ptr{Float64}, d_T::Ptr{Float64}, B::Integer;
    dim::Int=3, n_tau::Int, kF::Float64, beta::Float64, lambda::Float64, stream::Ptr{Cvoid}=C_NULL)
    a = [Int32.(v) for v in (leafType, leafOrder, leafInTau, leafOutTau, leafLoopIndex)]
    bs = Matrix{Float64}(loopbasis)
    GC.@preserve a bs begin
        tab = _FdgLeafTables(length(a[1]), size(bs, 2), size(bs, 1), dim, n_tau, pointer(a[1]), pointer(a[2]), pointer(a[3]),
            pointer(a[4]), pointer(a[5]), pointer(bs), kF, beta, lambda)
        _fdg_check(ccall((:fdg_leaf_eval_device, _libfdg), Cint,
            (Ref{_FdgLeafTables}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64, Ptr{Cvoid}),
            tab, d_K, 1, B, d_T, 1, B, d_leaf, 1, B, B, stream))
    end
    return nothing
end

"""
    leaf_eval_device_tiled!(d_leaf, leafType, ..., d_K, d_T, B; ...)

The same into a tile-major batch `Array{Float64,3}(64, L, cld(B, 64))` (`fdg_leaf_eval_device_tiled`): what `eval_device_tiled!` and
`accumulate_device_tiled!` read.
"""
function leaf_eval_device_tiled!(d_leaf::Ptr{Float64}, leafType::Vector{Int}, leafOrder::Vector{Int}, leafInTau::Vector{Int}, leafOutTau::Vector{Int},
    leafLoopIndex::Vector{Int}, loopbasis::Matrix{Float64}, d_K::Ptr{Float64}, d_T::Ptr{Float64}, B::Integer;
    dim::Int=3, n_tau::Int, kF::Float64, beta::Float64, lambda::Float64, stream::Ptr{Cvoid}=C_NULL)
    a = [Int32.(v) for v in (leafType, leafOrder, leafInTau, leafOutTau, leafLoopIndex)]
    bs = Matrix{Float64}(loopbasis)
    L = length(a[1])
    GC.@preserve a bs begin
        tab = _FdgLeafTables(L, size(bs, 2), size(bs, 1), dim, n_tau, pointer(a[1]), pointer(a[2]), pointer(a[3]),
            pointer(a[4]), pointer(a[5]), pointer(bs), kF, beta, lambda)
        _fdg_check(ccall((:fdg_leaf_eval_device_tiled, _libfdg), Cint,
            (Ref{_FdgLeafTables}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64, Int64, Int64, Ptr{Cvoid}),
            tab, d_K, 1, B, d_T, 1, B, d_leaf, 1, 64, 64 * L, B, stream))
    end
    return nothing
end
